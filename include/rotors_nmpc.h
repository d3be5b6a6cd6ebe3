/*
 * rotors_nmpc.h -- C ABI of the MI355X-native batched NMPC solver (librotors_nmpc_hip.so).
 *
 * Drop-in boundary: the object the reference binds at
 *   /root/reference/src/rotors_mpc_controller/controller.py:263   AcadosOcpSolver(ocp, json_file=...)
 * and drives at controller.py:412-460 through set()/solve()/get().  acados_template itself is
 * "Python -> ctypes -> C ABI of a generated shared object"; this header is the C ABI a
 * maintainer would bind instead (see INTEGRATION.md for the ctypes stub).
 *
 * Plain C: pointers and sizes only, no torch / HIP types in any signature (a stream is
 * passed as void*).  Every entry point names the reference call it replaces.
 *
 * Threading: one handle = one thread at a time (the reference serialises solve and rebuild
 * with _controller_lock, nodes/mpc_controller_node:122,193).  All calls are synchronous at
 * the ABI except nmpc_solve_batch_device, which only enqueues on the given stream.
 * One handle = one solve in flight: every solve uses the handle's single device workspace.  A caller that enqueues
 * nmpc_solve_batch_device on a stream of its own must synchronise that stream before the next call on the same handle that may
 * run elsewhere - another stream, or the host-buffer entry points nmpc_solve / nmpc_solve_batch, whose small-batch latency path
 * (B <= 64) runs on a private non-blocking stream (it waits for the NULL stream by itself, not for user streams).  Independent
 * batches in flight = independent handles (rotors_mpc_controller_amd/pipeline.py).
 */
#ifndef ROTORS_NMPC_H
#define ROTORS_NMPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NMPC_NX 13 /* controller.py:157 */
#define NMPC_NU 4  /* controller.py:158 */
#define NMPC_NY 17 /* controller.py:159 */

/* acados status numbering, consumed at controller.py:448 and mpc_controller_node:124 */
#define NMPC_SUCCESS 0
#define NMPC_NAN_DETECTED 1 /* a solve that failed AND an input of the instance (x0, yref, yref_e, x_init, u_init) is not finite */
#define NMPC_MAXITER 2
#define NMPC_MINSTEP 3
#define NMPC_QP_FAILURE 4   /* every other failed solve (min step, failed or untrusted factorisation, NaN from finite inputs) */

/* argument errors of set/get/solve_batch are negative and leave a message in nmpc_last_error */
#define NMPC_EARG (-1)
#define NMPC_EHIP (-2)

#define NMPC_DTYPE_F64 0   /* device buffers and arithmetic double */
#define NMPC_DTYPE_F32 1   /* since round 5 the same as NMPC_DTYPE_F32IO.  (Rounds 1-4: FP32 arithmetic on a kernel of its own, 5e-5 .. 5e-3 N
                              from the FP64 answer - narrower than the reference's own arithmetic (acados / HPIPM are double,
                              SURVEY 8) and outside the 1e-6 the path is held to: retired.)                                   */
#define NMPC_DTYPE_F32IO 2 /* device buffers float, arithmetic and workspace DOUBLE: the FP64 tile kernels read and write the
                              caller's float arrays directly (half the compulsory bytes of F64, u0 exact to float rounding of
                              the inputs).  Needs the team mapping without condensing (the default flags)                   */

/* flags */
#define NMPC_FLAG_SHARE_COLD_START 1u /* cold start: all stages share one (A,B,b); linearise once */
#define NMPC_FLAG_CONDENSED_QP 4u     /* QP phase as acados configures HPIPM for this reference: partial
                                         condensing to qp_cond_N blocks (controller.py:181,184) + IPM on the
                                         condensed QP.  Same solution (U8); fidelity / cross-check path, slow */
#define NMPC_FLAG_TEAM_MAPPING 2u     /* QP phase: 16 lanes cooperate on one instance (small batches)
                                         instead of one instance per lane (large batches) */

/* Everything controller.py:175-264 hands to AcadosOcp, plus the physical constants that the
 * reference bakes into the CasADi expression (controller.py:311-341), as plain numbers.   */
typedef struct nmpc_config {
    int32_t N;                 /* ocp.dims.N                         controller.py:179 */
    double dt;                 /* tf = N*dt                          controller.py:180 */
    double W[NMPC_NY];         /* diag(ocp.cost.W)                   controller.py:237-242 */
    double W_e[NMPC_NX];       /* diag(ocp.cost.W_e)                 controller.py:243 */
    double lbu[NMPC_NU];       /* ocp.constraints.lbu                controller.py:249 */
    double ubu[NMPC_NU];       /* ocp.constraints.ubu                controller.py:250 */
    double levenberg_marquardt;/* solver_options.levenberg_marquardt controller.py:190 */
    int32_t lm_scaled_by_dt;   /* [UPSTREAM U5] 1: stages use dt*lm (newer acados), 0: lm */
    int32_t cost_scaled_by_dt; /* [UPSTREAM U4] 1: stage cost times dt, terminal not */
    double mass;               /* controller.py:73  */
    double gravity;            /* controller.py:74  */
    double inertia[3];         /* diagonal only is used, controller.py:81-83 */
    double rotor_x[NMPC_NU];   /* controller.py:100 */
    double rotor_y[NMPC_NU];   /* controller.py:101 */
    double rotor_z[NMPC_NU];   /* spin*k_m, controller.py:103 */
    int32_t sim_num_stages;    /* must be 2 (explicit midpoint)      controller.py:187 */
    int32_t sim_num_steps;     /* controller.py:188 */
    int32_t qp_iter_max;       /* qp_solver_iter_max                 controller.py:185 */
    int32_t qp_cond_N;         /* qp_solver_cond_N                   controller.py:184 (solution-invariant, U8) */
    double qp_tol_comp;        /* IPM: stop when mu <= tol_comp and ... */
    double qp_tol_stat;        /* ... relative stationarity factor <= tol_stat */
    double qp_mu0;             /* initial barrier parameter */
    double qp_tau;             /* fraction to the boundary */
    double qp_thr0;            /* initial distance from the bounds, absolute ... */
    double qp_thr0_rel;        /* ... and relative to the box width (the larger applies) */
    int32_t dtype;             /* NMPC_DTYPE_F64 | NMPC_DTYPE_F32IO (NMPC_DTYPE_F32 is accepted as the latter) */
    int32_t device;            /* HIP device ordinal */
    int32_t max_batch;         /* workspace is sized for this many instances */
    uint32_t flags;            /* NMPC_FLAG_* */
    /* Active-set polish of the QP (team mapping only; not in HPIPM): pin the bounds the iterate marks
     * active, solve the remaining LQ problem exactly, accept only if the KKT conditions of the QP hold
     * (then the result is the exact QP solution); otherwise correct the active set (primal-dual
     * active-set step) or fall back to the interior point iteration.                                  */
    int32_t qp_polish;         /* 0 = plain IPM, 1 = on */
    int32_t qp_polish_passes;  /* active-set corrections per attempt; 0 (default) = the measured policy for the horizon: 8 below N = 160, 16 from there up */
    int32_t qp_polish_budget;  /* no new attempt after this many passes; 0 (default) = 16: two attempts on short horizons, ONE on long ones */
    double qp_polish_mu;       /* first attempt when mu <= this (>= mu0: before any IPM iteration), then every 100x below */
    int32_t qp_polish_ckpt;    /* leading stages whose Riccati state (P_k, p_k) a corrected active-set pass checkpoints
                                  (the first pass of an attempt keeps at most two): the next pass refactorises only
                                  stages <= the highest stage whose pin set changed when a current checkpoint
                                  covers it, else the whole horizon.  0 = always the whole horizon                     */
    int32_t qp_maxiter_status; /* [UPSTREAM U10] status of a solve whose QP hits qp_iter_max: 0 = tolerated (current acados SQP_RTI),
                                  2 = NMPC_MAXITER is returned (some acados versions; the caller then discards the command and
                                  cold-starts, controller.py:448-450, nodes/mpc_controller_node:124)                           */
    /* Accuracy certificate of the Riccati factorisations (FP64 tile kernels; this build's own device, not HPIPM's).
     * g = max_k |B_k' P_{k+1} B_k| of a backward sweep, against g of the FIRST factorisation of the solve: pinned inputs
     * (active-set pass) or heavily penalised ones (late interior-point iterations) leave stretches of the horizon open loop, on
     * an unstable plant P then grows by rho(A)^2 per stage and the recursion loses about 1e-15 * growth of relative accuracy.
     * A sweep with g > qp_growth_max * g_first is not trusted: an active-set pass is not accepted (no further attempt), an
     * interior-point iteration is not taken - the QP ends at its iterate: status 0 if mu <= qp_acc_comp and the stationarity
     * factor <= qp_acc_stat, NMPC_QP_FAILURE otherwise.  0 = certificate off.  Never reached with the reference's vehicle.   */
    double qp_growth_max;      /* default 1e6 */
    double qp_acc_comp;        /* default 1e-8  ([UPSTREAM] HPIPM's default complementarity tolerance) */
    double qp_acc_stat;        /* default 1e-8 */
    double qp_tol_step;        /* convergence also needs the last step max |alpha d| / (ubu - lbu) <= this; default 1e-3, 0 = off */
    int32_t qp_warm_start;     /* 1 (default): an active-set attempt that runs out of passes hands its last pass to the interior-point
                                  iteration as the start point (inputs 1e-3 of the box inside the bounds, multipliers from the pass,
                                  floored at mu = 1e-3) instead of the cold start; the next attempt then follows after one iteration   */
    int32_t reserved_;
} nmpc_config;

typedef struct nmpc_stats {
    int32_t batch;             /* instances in the last solve */
    int32_t iter_min, iter_max;/* IPM iterations over the batch */
    double iter_mean;
    int32_t n_status[5];       /* histogram of the acados status codes */
    double ms_prepare;         /* device time of the linearisation kernel (HIP events); 0 when it is fused into the solve */
    double ms_solve;           /* device time of the solve kernel; both 0 after nmpc_set_timing(s, 0) */
    uint64_t workspace_bytes;
    double polish_mean;        /* active-set passes per instance (team mapping), mean / max */
    int32_t polish_max;
    int32_t n_polished;        /* instances that finished with an accepted active-set solution */
    int32_t n_tail;            /* instances whose first active-set attempt failed (continued on the interior point; default path) */
    double ms_tail;            /* device time of the launches after k_team_as (k_team_qp_list / the block-parallel tail), 0 when k_team_as continued them itself */
} nmpc_stats;

typedef struct nmpc_solver nmpc_solver; /* opaque; owns all device memory */

/* values of reference config/params.yaml + the acados defaults the reference relies on */
void nmpc_default_config(nmpc_config *cfg);

/* replaces AcadosOcpSolver(ocp, json_file=...)  controller.py:263 -- no codegen, no compile,
 * no on-disk artefacts.  NULL on failure (message via nmpc_last_error(NULL)).              */
nmpc_solver *nmpc_create(const nmpc_config *cfg);

/* replaces `del old_solver` + rmtree of the codegen dir  controller.py:169-172 */
void nmpc_destroy(nmpc_solver *s);

/* replaces AcadosOcpSolver.set(stage, field, value)  controller.py:414-445.
 * fields: "x" (n=13, stage 0..N), "u" (n=4, 0..N-1), "yref" (n=17 for stage<N, 13 at N),
 * "lbx"/"ubx" (n=13, stage 0 only: the initial-state pin).  Single-instance slot.        */
int nmpc_set(nmpc_solver *s, int stage, const char *field, const double *value, int n);

/* replaces AcadosOcpSolver.get(stage, field)  controller.py:452-460; fields "x", "u" */
int nmpc_get(nmpc_solver *s, int stage, const char *field, double *out, int n);

/* replaces AcadosOcpSolver.solve()  controller.py:447: one SQP real-time iteration on the
 * single-instance slot, run on the GPU (batch of one).  Returns the acados status.         */
int nmpc_solve(nmpc_solver *s);

/* Batched RTI, host buffers (copied in and out; PCIe inclusive).  All arrays are row-major
 * doubles regardless of cfg.dtype.
 *   x0      [B][13]
 *   yref    [B][N][17] or, if yref_bcast, [N][17] shared by all instances
 *   yref_e  [B][13]    or [13]
 *   x_init  [B][N+1][13], u_init [B][N][4]: linearisation point (warm start,
 *           controller.py:419-424); both NULL = cold start x_k = x0, u_k = 0 (:425-431)
 *   u0      [B][4] out (zeros where status != 0, controller.py:448-450), status [B] out
 *   x_out   [B][N+1][13], u_out [B][N][4] out, nullable
 * Returns 0 or a negative argument/HIP error.                                             */
int nmpc_solve_batch(nmpc_solver *s, int B, const double *x0, const double *yref,
                     const double *yref_e, int yref_bcast, const double *x_init,
                     const double *u_init, double *u0, int32_t *status, double *x_out,
                     double *u_out);

/* Same, all pointers are DEVICE pointers of the buffer element type of cfg.dtype (double; float for F32 / F32IO) and
 * the work is only enqueued on `hip_stream` (a hipStream_t passed as void*; NULL = default).
 * status is int32 on the device.  This is the entry the benchmark times.                   */
int nmpc_solve_batch_device(nmpc_solver *s, int B, const void *x0, const void *yref,
                            const void *yref_e, int yref_bcast, const void *x_init,
                            const void *u_init, void *u0, int32_t *status, void *x_out,
                            void *u_out, void *hip_stream);

/* device-side IPM iteration counts of the last solve: int32 [B] DEVICE pointer (read-only) */
const int32_t *nmpc_device_iterations(nmpc_solver *s);

/* device-side active-set pass counts of the last solve (team mapping): int32 [B] DEVICE pointer (read-only);
 * > 0: the instance ended on an accepted active-set solution after that many passes, <= 0: it did not
 * (interior-point result), the magnitude being the passes spent                                          */
const int32_t *nmpc_device_passes(nmpc_solver *s);

/* synchronises, then copies the two arrays above to the host: iterations [n], passes [n] (either may be NULL), n <= max_batch */
int nmpc_get_counts(nmpc_solver *s, int n, int32_t *iterations, int32_t *passes);

/* synchronises, then fills iteration / status histograms and kernel times of the last solve */
int nmpc_get_stats(nmpc_solver *s, nmpc_stats *out);

/* on (default): every solve is bracketed by HIP events on its stream and nmpc_get_stats reports the
 * kernel times; off: no events are recorded (two fewer stream operations per solve) and the times read 0 */
int nmpc_set_timing(nmpc_solver *s, int on);

/* message of the last failing call on this handle (s == NULL: of the last failed create) */
const char *nmpc_last_error(const nmpc_solver *s);

/* ---- the steps either side of the solve, on the device; pointers are DEVICE pointers of
 * element type cfg.dtype, work is enqueued on `hip_stream` ------------------------------------- */

/* replaces ReferenceGenerator.build_horizon (reference.py:75-91) + the yref stacking of
 * controller.py:433-445 for B constant setpoints: positions [B][3], yaws [B] (quaternion from yaw as
 * reference.py:11-13), thrust reference per motor -> yref [B][N][17], yref_e [B][13]             */
int nmpc_build_hover_reference_device(nmpc_solver *s, int B, const void *positions, const void *yaws,
                                      double thrust_per_motor, void *yref, void *yref_e, void *hip_stream);

/* replaces MPCControllerNode._odom_cb (nodes/mpc_controller_node:88-113): pose [B][7] = position(3),
 * orientation (x,y,z,w); twist [B][6] = body-frame linear(3), angular(3) -> x0 [B][13]            */
int nmpc_odometry_to_state_device(nmpc_solver *s, int B, const void *pose, const void *twist, void *x0,
                                  void *hip_stream);

/* replaces MPCControllerNode._publish_command (nodes/mpc_controller_node:152-164): u [B][4] thrusts ->
 * motor speeds [B][4] (rad/s); `clipped` [B][4] (nullable) receives the bound-clipped thrusts      */
int nmpc_commands_to_motor_speeds_device(nmpc_solver *s, int B, const void *u, double rotor_force_constant,
                                         double motor_min_speed, double motor_max_speed, void *speeds,
                                         void *clipped, void *hip_stream);

/* replaces the command hand-over of MPCControllerNode._loop (nodes/mpc_controller_node:122-131): where
 * status [B] is 0 the command u0 [B][4], clipped to the input bounds (:153-154), becomes the held command
 * (_last_command, :164); elsewhere the previous held command stays (re-published, :126-127).
 * held [B][4] is read and written; the caller initialises it (the node has no command before its first
 * successful solve).                                                                                   */
int nmpc_hold_command_device(nmpc_solver *s, int B, const void *u0, const int32_t *status, void *held,
                             void *hip_stream);

/* plant step of a closed-loop rollout: x [B][13], u [B][4] -> x_next [B][13] with the
 * controller's own model and ERK scheme (controller.py:183-188,267-355) over one interval dt;
 * normalize_q != 0 renormalises the quaternion as controller.py:406-409 does on every tick          */
int nmpc_plant_step_device(nmpc_solver *s, int B, const void *x, const void *u, void *x_next,
                           int normalize_q, void *hip_stream);

/* one closed-loop tick's plant side in a single launch: nmpc_hold_command_device followed by nmpc_plant_step_device
 * of the held command, the next measured state written over x [B][13] (nodes/mpc_controller_node:122-131,152-164 and
 * the plant of SURVEY 8f-2).  Same results as the two calls; exists because a tick of 4096 vehicles is launch-bound */
int nmpc_hold_and_step_device(nmpc_solver *s, int B, const void *u0, const int32_t *status, void *held, void *x,
                              int normalize_q, void *hip_stream);

/* adjoint sensitivities of the model (the CasADi-generated `expl_vde_adj` of controller.py:267-355, which
 * acados compiles next to expl_vde_forw).  x [B][13], u [B][4], lam [B][13] -> out [B][17]:
 *   continuous != 0:  ( f_x(x,u)' lam | f_u(x,u)' lam )          -- what expl_vde_adj evaluates
 *   continuous == 0:  ( A' lam | B' lam ) of the shooting interval that starts at (x,u), by the reverse sweep through the
 *                     ERK scheme of controller.py:183-188 (no A, B formed): the discrete adjoint of expl_vde_forw's result */
int nmpc_adjoint_sensitivities_device(nmpc_solver *s, int B, const void *x, const void *u, const void *lam, void *out,
                                      int continuous, void *hip_stream);

/* stationarity / feasibility report of trajectories (e.g. the x_out / u_out of a solve) for the NLP of controller.py:175-264,
 * one adjoint sweep per instance: res [B][3] = ( max |projected Lagrangian gradient w.r.t. the inputs|,
 * max dynamics defect |phi(x_k,u_k) - x_{k+1}|, max input-bound violation ).  acados reports the same quantities as
 * res_stat / res_eq / res_ineq of its SQP; after ONE real-time iteration they measure how far the step is from converged */
int nmpc_kkt_report_device(nmpc_solver *s, int B, const void *x_traj, const void *u_traj, const void *yref, const void *yref_e,
                           int yref_bcast, void *res, void *hip_stream);

/* Parallel-in-time Riccati factorisation of ONE active-set pass (controller.py:184 asks HPIPM for partial condensing
 * into qp_solver_cond_N = min(N, 5) blocks; cfg/rotors_mpc.cfg:9 lets the horizon reach 600).  The horizon is cut into `blocks`
 * blocks that are swept by their own teams AT THE SAME TIME, with `blocks - 1` sequential boundary updates in between
 * (csrc/nmpc_block.hpp has the algebra) - the blocks of the reference's condensing, used as parallelism instead of as 480-input
 * dense stages.  A building block with its own entry point; nmpc_solve_batch* uses the same kernels by itself from N = 160 up
 * (the block-parallel tail of long-horizon solves).
 *   The LQ problem factorised is the one the LAST solve of this handle ended on - its per-stage linearisation (so that solve must
 *   have been warm-started, or NMPC_FLAG_SHARE_COLD_START off) and the pin set its last forward sweep left; x0 .. u_init are the
 *   device arrays that solve was given.  FP64 arithmetic in the team mapping only.
 *   factors_out  [B][N][80]  doubles, or NULL: per stage Mbar' as 4 tiles x 16 lanes | the L^-1 tile (the layout of the solver's own sweeps)
 *   boundary_out [B][J+1][256] doubles, or NULL: value function Pbar (16 tiles x 16 lanes) at the start of block j
 *   check_out    [B][J+1][256] doubles, or NULL: the same as recomputed by block j's own final sweep (blocks 0 .. J-2)
 *   ms_out       float[3], or NULL: device time of the three launches (synchronises)
 * blocks = 1 is the sequential sweep in the same code.  Returns J, the number of blocks that hold stages (<= blocks), or < 0. */
int nmpc_block_factor_device(nmpc_solver *s, int B, int blocks, const void *x0, const void *yref, const void *yref_e, int yref_bcast,
                             const void *x_init, const void *u_init, double *factors_out, double *boundary_out, double *check_out,
                             float *ms_out, void *hip_stream);

/* diagnostic: the factors the last solve's own sweeps left in the workspace, to the HOST [B][N][80] (same layout as factors_out;
 * complete only for a solve that ran without the LDS stage cache, NMPC_TEAM_LSTG=0) */
int nmpc_debug_factors(nmpc_solver *s, int B, double *host_out);

/* diagnostic: where the block-parallel tail of long-horizon solves (csrc/nmpc_block.hpp) left the instances
 * of the last solve: host_out [B] = 0 not in the work list, 3 finished by the tail, 5 handed on to the sequential work-list kernel.
 * Returns the number of blocks the tail cuts the horizon into (0: this handle runs no tail - N < 160 unless NMPC_BLOCK_TAIL=1) */
int nmpc_debug_tail_states(nmpc_solver *s, int B, int32_t *host_out);

/* diagnostic: canary bands around every device buffer of the handle.  A handle created with NMPC_GUARD=<KiB> in the environment
 * places each of its allocations between two bands of that size filled with a pattern; this call synchronises the device and
 * returns the number of band bytes that no longer hold it (0 = no store outside any buffer within that distance; nmpc_last_error
 * names the first damaged buffer), -1 when the handle was created without NMPC_GUARD.  A store past a buffer that stays inside
 * mapped memory does not fault - this is how it shows (tests/test_gpu_parity.py runs the 3-4 integrator-step regression under it). */
long long nmpc_debug_guard_check(nmpc_solver *s);

/* diagnostic: the launch schedule of the last solve, as bits - 1: first attempt by k_team_as (the default FP64 path; 0: one general kernel),
 * 2: failed first attempts were continued inside k_team_as (no work-list launch), 4: the block-parallel tail of a long horizon ran.
 * bench.py names the kernels of its roofline object from it.  Negative: an error code. */
int nmpc_debug_last_schedule(const nmpc_solver *s);

/* "rotors_nmpc_hip <abi> (gfx950) src <sha1[:12] of the kernel sources the binary was built from>" */
const char *nmpc_version(void);

/* sizeof(nmpc_config) and sizeof(nmpc_stats) of THIS binary (either pointer may be NULL); returns the ABI revision (3).  A
 * binding compares them with its own mirror of the structs before the first nmpc_create / nmpc_get_stats: both structs grew in
 * rounds 2 and 3, and nmpc_get_stats writes sizeof(nmpc_stats) bytes into the caller's buffer.                               */
int nmpc_abi_sizes(int *config_bytes, int *stats_bytes);

#ifdef __cplusplus
}
#endif
#endif /* ROTORS_NMPC_H */
