/*
 * nmpc_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C FP64 restatement of the hot path that the reference reaches through
 * `AcadosOcpSolver.solve()` (reference: src/rotors_mpc_controller/controller.py:447):
 * one SQP real-time iteration on the rotor-level quadrotor OCP defined in
 * controller.py:175-355.
 *
 * PARITY STATUS: "parity unpinned" against acados itself.  acados / HPIPM / BLASFEO /
 * CasADi are un-vendored third-party dependencies of the reference, absent from
 * /root/reference and from this image, and the reference pins no version of them and
 * ships no tests or golden vectors.  The oracle therefore restates the *published*
 * algorithms (explicit-midpoint ERK with forward variational equations, Gauss-Newton
 * LINEAR_LS QP, Levenberg-Marquardt term, Mehrotra predictor-corrector IPM on the
 * Riccati-factorised OCP-QP, HPIPM-style partial condensing) and is pinned by analytic
 * known-answer tests and independent cross-checks (tests/test_oracle_*.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
 */
#ifndef NMPC_ORACLE_H
#define NMPC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NX 13
#define ORC_NU 4
#define ORC_NY 17

typedef struct orc_config {
    int N;                  /* horizon_steps          controller.py:179 */
    double dt;              /* step, tf = N*dt        controller.py:180 */
    double W[ORC_NY];       /* diag of stage weight   controller.py:237-242 */
    double We[ORC_NX];      /* diag of terminal wt.   controller.py:243 */
    double lbu[ORC_NU];     /* controller.py:249 */
    double ubu[ORC_NU];     /* controller.py:250 */
    double lm;              /* levenberg_marquardt    controller.py:190 */
    int lm_scaled_by_dt;    /* [UPSTREAM] U5 switch: 1 = stages use dt*lm, terminal lm */
    int cost_scaled_by_dt;  /* [UPSTREAM] U4 switch: 1 = stage cost times dt, terminal not */
    double mass;            /* controller.py:73 */
    double gravity;         /* controller.py:74 */
    double J[3];            /* inertia diagonal       controller.py:81-83 */
    double rotor_x[ORC_NU]; /* controller.py:100 */
    double rotor_y[ORC_NU]; /* controller.py:101 */
    double rotor_z[ORC_NU]; /* spin * k_m             controller.py:103 */
    int sim_num_stages;     /* controller.py:187 (2 = explicit midpoint) */
    int sim_num_steps;      /* controller.py:188 */
    int qp_iter_max;        /* controller.py:185 */
    int qp_cond_N;          /* controller.py:184; 0 or >=N = solve uncondensed */
    double qp_tol_comp;     /* IPM: stop when mu <= tol_comp ... */
    double qp_tol_stat;     /* ... and relative stationarity reduction <= tol_stat */
    double qp_mu0;          /* IPM initial barrier parameter */
    double qp_tau;          /* fraction to the boundary */
    double qp_thr0;         /* initial distance from the bounds (absolute) */
    double qp_thr0_rel;     /* ... and relative to the box width; the larger applies */
    double qp_gamma;        /* centrality safeguard: products >= gamma*mu after a step; 0 = off */
    int qp_polish;          /* 1: active-set polish once mu <= qp_polish_mu (then every 100x below) */
    double qp_polish_mu;
    int qp_polish_passes;   /* primal-dual active-set corrections per polish attempt */
    int qp_polish_budget;   /* no further attempt once this many passes were spent */
    /* accuracy certificate of the Riccati factorisations (this build's own device, not HPIPM's): g = max_k |B_k' P_{k+1} B_k|
     * of a backward sweep against g of the FIRST factorisation of the solve.  Pinned (active-set pass) or heavily penalised
     * (late interior-point iterations) inputs leave stretches of the horizon open loop; on an unstable plant P then grows by
     * rho(A)^2 per stage and the recursion loses ~1e-15 * growth of relative accuracy.  A sweep whose g exceeds
     * qp_growth_max * g_first is not trusted: an active-set pass is not accepted (and no further attempt is made), an
     * interior-point iteration is not taken - the QP then ends at its current iterate, status 0 if that iterate is within the
     * acceptable tolerances below and QP failure otherwise.  0 = certificate off.                                           */
    double qp_growth_max;
    double qp_acc_comp;     /* acceptable level: mu <= qp_acc_comp and rho <= qp_acc_stat (HPIPM's default tolerances) */
    double qp_acc_stat;
    double qp_tol_step;     /* convergence also needs the last step max|alpha d| / (ub - lb) <= this; 0 = off */
    int qp_maxiter_status;  /* [UPSTREAM] U10 switch: status returned when the QP hits qp_iter_max: 0 = tolerated (current
                               acados SQP_RTI), 2 = reported (some versions; the caller then discards the command,
                               controller.py:448-450) */
    int qp_warm_start;      /* 1: an attempt that runs out of passes hands its last pass to the interior point as the start point */
    int qp_exit_mode;       /* 0 (default): exit on the tracked measures - mean complementarity mu <= qp_tol_comp, stationarity factor
                               rho <= qp_tol_stat, last step <= qp_tol_step.  1: [UPSTREAM U9] HPIPM's own exit test - the TRUE
                               residuals of the iterate in the infinity norm, stationarity and bound equations <= qp_tol_stat, max
                               complementarity product <= qp_tol_comp (HPIPM / acados default 1e-8 each; controller.py:179-190 sets
                               none).  Mode 1 exists to predict acados' accuracy floor on this OCP (tests/test_acados_floor.py)   */
    int qp_bound_res;       /* 1: the residuals of the bound equations, u - lo - t_l and hi - u - t_u (HPIPM's res_d), are fed back into the
                               Newton system of every interior-point iteration.  0 (default, what the kernels do): left out - the start is
                               feasible, t follows u exactly in exact arithmetic, what the residuals hold is rounding of size ulp(u)      */
} orc_config;

typedef struct orc_stats {
    int qp_iter;            /* IPM iterations taken */
    int qp_status;          /* 0 ok, 2 max iter, 3 min step / factorisation failure, 4 growth certificate (untrusted), 1 nan */
    double res_stat;        /* TRUE inf-norm stationarity residual at exit (recomputed) */
    double res_eq;          /* TRUE inf-norm dynamics residual at exit */
    double res_comp;        /* TRUE max complementarity product at exit */
    double mu;              /* barrier parameter at exit */
    double rho;             /* tracked relative stationarity factor prod(1-alpha) */
    int hess_projected;     /* PROJECT_REDUC_HESS had to act (expected 0) */
    int polished;           /* the active-set polish was accepted */
    int polish_attempts;
    double growth;          /* largest g / g_first seen in a factorisation of this solve */
    double step_last;       /* max |alpha d| / (ub - lb) of the last interior-point step */
    int untrusted;          /* 1: the growth certificate stopped an attempt or the iteration */
} orc_stats;

/* defaults = reference config/params.yaml + acados option defaults */
void orc_default_config(orc_config *c);

/* model: controller.py:267-355 */
void orc_model_f(const orc_config *c, const double *x, const double *u, double *f);
void orc_model_jac(const orc_config *c, const double *x, const double *u,
                   double *fx /*13x13 row-major*/, double *fu /*13x4 row-major*/);
/* counterparts of the CasADi-generated functions */
void orc_vde_forw(const orc_config *c, const double *x, const double *Sx, const double *Su,
                  const double *u, double *xdot, double *Sxdot, double *Sudot);
void orc_vde_adj(const orc_config *c, const double *x, const double *lam, const double *u,
                 double *adj /*17 = [fx' lam ; fu' lam]*/);

/* ERK over one shooting interval with forward sensitivities: xn = phi(x,u), A, B */
void orc_integrate(const orc_config *c, const double *x, const double *u,
                   double *xn, double *A /*13x13*/, double *B /*13x4*/);

/* linearisation of the whole horizon (SQP_RTI preparation phase).
 * outputs (row-major, stage-major): A[N][13][13], B[N][13][4], b[N][13],
 * q[N+1][13], r[N][4], lo[N][4], hi[N][4], Qd[N+1][13], Rd[N][4]                     */
void orc_linearize(const orc_config *c, const double *xtraj, const double *utraj,
                   const double *yref, const double *yref_e,
                   double *A, double *B, double *b, double *q, double *r,
                   double *lo, double *hi, double *Qd, double *Rd, int *hess_projected);

/* QP: Riccati-based Mehrotra IPM on the (optionally partially condensed) OCP-QP.
 * dx[N+1][13], du[N][4] out.  returns qp status                                       */
int orc_qp_solve(const orc_config *c, const double *dx0,
                 const double *A, const double *B, const double *b,
                 const double *q, const double *r, const double *lo, const double *hi,
                 const double *Qd, const double *Rd,
                 double *dx, double *du, orc_stats *st);

/* one SQP real-time iteration. xtraj[(N+1)*13], utraj[N*4] are in/out (init -> result).
 * yref[N*17], yref_e[13]. returns acados-style status (0 ok, 1 nan, 4 qp failure)     */
int orc_sqp_rti(const orc_config *c, const double *x0, const double *yref,
                const double *yref_e, double *xtraj, double *utraj, orc_stats *st);

/* batch driver, cold start as controller.py:425-431 (x_k = x0, u_k = 0) unless
 * x_init/u_init given.  yref_bcast != 0: yref is [N][17], yref_e [13] shared by all.
 * nthreads <= 0: all OpenMP threads.                                                  */
int orc_solve_batch(const orc_config *c, int B, const double *x0, const double *yref,
                    const double *yref_e, int yref_bcast,
                    const double *x_init, const double *u_init,
                    double *u0, int *status, double *x_out, double *u_out,
                    int *iters, int nthreads);

/* the same with two more per-instance outputs (nullable): passes = active-set passes spent, > 0 when the instance ended on an
 * accepted active-set solution, <= 0 when on the interior-point iterate (the library's nmpc_device_passes convention);
 * growth = largest g / g_first of the growth certificate                                                                     */
int orc_solve_batch_ex(const orc_config *c, int B, const double *x0, const double *yref,
                       const double *yref_e, int yref_bcast,
                       const double *x_init, const double *u_init,
                       double *u0, int *status, double *x_out, double *u_out,
                       int *iters, int *passes, double *growth, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
