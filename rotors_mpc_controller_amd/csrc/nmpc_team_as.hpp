// nmpc_team_as.hpp -- the solver kernels of the team mapping (gfx950 device code): ONE source, four modes.
//
// team_as<MODE>:  0  preparation (RTI linearisation, controller.py:419-445 staging) + the FIRST active-set attempt of the QP (up to
//                    qp_polish_passes passes of "pin the active bounds, solve the remaining LQ problem with one Riccati factorisation +
//                    forward sweep, check the KKT conditions of the QP in the same sweep")                        -> k_team_as
//                 1  the whole QP of every instance of the batch: interior-point iterations, attempts in between   -> k_team_qp
//                 2  the same for work-list instances, continuing after a failed first attempt                     -> k_team_qp_list, and
//                    inlined behind MODE 0 in k_team_as (team_as_kernel: a failed attempt continues on its own wave, one launch per solve)
//                 3  one step of the block-parallel tail of long horizons (no factor sweep: nmpc_block.hpp did it)  -> k_team_tail
// Four instances per wave: team b = lanes {16a + 4b + c}, lane (a,c) = element (a,c) of every 4 x 4 register tile, all products on
// v_mfma_f64_4x4x4_4b.  On the near-hover set 99.3 % of the instances are accepted after the first pass and all of them within three.
//
// Why the first attempt is code of its own (MODE 0): it carries no interior-point state, takes its
// per-lane constants into vector registers once (the 200-dword constant block is not touched inside a
// sweep), stages nothing it can form on the fly (the cost gradients come straight from yref / x_init: no
// qr array, no xl / ul copy) and needs 10 KB of LDS per wave, so two waves fit a SIMD.
//
// Arithmetic: the oracle restates the algorithm (oracle/nmpc_oracle.c: active-set polish, ocpqp_ipm) and the GPU parity tests hold
// 1e-9 against it; pass and iteration counts are the oracle's.
//
// Linearisation: the state of a shooting interval is integrated by ONE lane (lane r of a team takes stage
// k0 + r of a chunk of AS_CH stages: 4 model evaluations per stage instead of 4 replicated in 16 lanes),
// which leaves the points at which the model Jacobian is needed -- q, omega at the start and at the
// midpoint of every ERK step -- in LDS; the forward sensitivities are then propagated per stage in tile form
// (12 MFMAs per variational-equation evaluation, Jacobian tiles from per-lane coefficient patterns).
#pragma once

#include <type_traits>

#include "nmpc_ipm.hpp"
#include "nmpc_team.hpp"
#include "nmpc_stage.hpp"

namespace nmpc {

constexpr int AS_CH = 8;             // stages linearised per chunk (per-stage variant); lanes r < AS_CH integrate
constexpr int AS_EV = 56;            // doubles per stage in the evaluation-point buffer: 2 steps x 2 points x 7 | t2 | b_k | R e3 / m at the 4 points | 0
constexpr int AS_MAX_STEPS = 4;      // sim_method_num_steps these kernels are built for (controller.py:188 sets 2).  More than two
                                     // integrator steps take a second layout of the evaluation-point buffer - twice the room per
                                     // stage, half the stages per chunk, so the LDS carve is the same (the shared variant's single
                                     // stage takes 56 doubles more: the host places the stage cache accordingly, lm_off)
// layout of one stage's evaluation points for at most MS integrator steps: MS x 2 points x 7 | t2 | b_k (13) | R e3 / m at the 2 MS points | 0
template <int MS> struct EvLayout {
    static constexpr int T2 = MS * 14, Bk = T2 + 1, RE = Bk + 13, ZERO = RE + MS * 6;
    static constexpr int STRIDE = MS <= 2 ? AS_EV : 2 * AS_EV;      // 55 of 56, 95 of 112 doubles used
    static constexpr int CHUNK = MS <= 2 ? 8 : 4;                   // stages linearised per chunk (per-stage variant)
};
constexpr int AS_LM_ROWS = 80;       // doubles per stage in the LDS stage cache: Mbar^T tiles (64) | L^-1 tile (16)
constexpr int IP_LM_ROWS = 88;       // ... of the kernels that also iterate the interior point method: | 1 / d_a of H_uu = L D L' (4) | pad
// LDS carve per team, in doubles
constexpr int A_AD = 0;              // [16][8]  rows of the dense A columns (natural layout)
constexpr int A_B = A_AD + 128;      // [16][4]
constexpr int A_BV = A_B + 64;       // [16]
constexpr int A_HG = A_BV + 16;      // [16]     Huu (10) | gu (4)
constexpr int A_D = A_HG + 16;       // [16]     D | rhat | free mask | pinned value
constexpr int A_H = A_D + 16;        // [16]     stage gradient, natural rows
constexpr int A_XH = A_H + 16;       // [16]
constexpr int A_RED = A_XH + 16;     // [32]     small reductions
constexpr int A_EV = A_RED + 40;     // evaluation points of the linearisation
// (the host pads the team stride - carve + stage cache - to 192 B past a multiple of the 256-B bank row: see TEAM_LDS in nmpc_team.hpp)
constexpr int TEAM_AS_LDS_SHARED = A_EV + AS_EV;            // 368
constexpr int TEAM_AS_LDS_STAGE = A_EV + AS_CH * AS_EV;     // 760

// Which builds of k_team_as carry the continuation of failed first attempts (team_as_kernel below; host and device agree through this one
// function): all of the one-wave builds.  A switch per variant because the translation units that run are built with
// -amdgpu-sched-strategy=iterative-ilp, and that strategy has produced ONE kernel - the per-stage build without trajectories, with the
// continuation inlined - in which a register-parking copy sat in front of the EXEC restore of a join block.
// tools/emu/exec_join_check.py finds the pattern and tests/test_isa_emulation.py runs it on every flag build; that kernel compiled clean
// once jac_coef's selects stopped being an if / else (nmpc_team.hpp).  Should the check ever fail again for a variant, returning false for
// it here gives it back the work-list launch, whose code is unchanged.
constexpr bool as_cont_built(bool shared, bool traj) { (void)shared; (void)traj; return true; }

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)

// transposed Jacobian tiles at an evaluation point given as (q, omega, t2 = 2 (sum u) / m); re3 = (R e3 / m, 0): the thrust
// direction column, quadratic in q, comes from the lane that integrated the stage (thrust_dir below) - one LDS read here
// instead of ten FP64 operations and three selects per point in every lane
__device__ __forceinline__ void jac_tiles_ev(const JacCoef<double> &k, const double *e, double re3a, double t2,
                                             double &FvqT, double &FqqT, double &FqwT, double &FwwT, double &r3a)
{
    const double qw = e[0], qx = e[1], qy = e[2], qz = e[3], wx = e[4], wy = e[5], wz = e[6];
    FvqT = t2 * (k.vq[0] * qw + k.vq[1] * qx + k.vq[2] * qy + k.vq[3] * qz);
    FqqT = k.qq[0] * wx + k.qq[1] * wy + k.qq[2] * wz;
    FqwT = k.qw[0] * qw + k.qw[1] * qx + k.qw[2] * qy + k.qw[3] * qz;
    FwwT = k.ww[0] * wx + k.ww[1] * wy + k.ww[2] * wz;
    r3a = re3a;
}
__device__ __forceinline__ void thrust_dir(double inv_mass, const double *x, double *out)
{
    const double qw = x[6], qx = x[7], qy = x[8], qz = x[9];
    out[0] = 2.0 * (qx * qz + qw * qy) * inv_mass;
    out[1] = 2.0 * (qy * qz - qw * qx) * inv_mass;
    out[2] = (1.0 - 2.0 * (qx * qx + qy * qy)) * inv_mass;
}

// work list of the instances the active-set kernel hands to the general kernel
struct WorkList {
    int *count;     // [1] entries appended by this launch (zero on entry; reset by the consumer)
    int *done;      // [1] consumer workgroups finished (the last one resets both)
    int *list;      // [Bp] instance indices
};

// context of one step of the block-parallel tail (MODE 3)
struct TailCtx {
    double *ts = nullptr;            // [Bp + 1][TS_ROWS] tail state (nmpc_team.hpp)
    const double *binfo = nullptr;   // [Bp + 1][J][2] verdict of the block factorisation of this step (nmpc_block.hpp)
    int J = 0;
    int phase = 0;                   // 0: install the warm start | 1: the rest of an interior-point iteration | 2: the rest of an active-set pass
                                     // 3: forward sweep of ONE block of an active-set pass (then phase 2 only decides)
    int *fb_count = nullptr;         // fallback list: what the tail does not finish (solved by k_team_qp_list from the hand-over)
    int *fb_list = nullptr;
    int cap = 0;                     // MODE 0: passes the first launch performs before it hands a running attempt to the tail (0: all)
    int inplace = 0;                 // MODE 0: a team whose first attempt fails is NOT appended to the work list - team_as returns true for its
                                     // lanes and the kernel continues it at once (MODE 2 on the same wave: no second launch)
    double *frec = nullptr;          // [Bp + 1][J][FR_ROWS] records of the block-parallel forward sweep (phase 3 writes, phase 2 reads), or null
    const double *xb = nullptr;      // [Bp + 1][J][16] state at the start of every block (left by the boundary scan)
    int blk = 0, M = 0;              // phase 3: this team's block, stages per block
    int *nx_count = nullptr;         // the work list of the NEXT step: instances that are still in the tail after this one (the list is
    int *nx_list = nullptr;          // compacted from step to step: a wave costs the same with one live team as with four)
};

// MODE 0: preparation + the FIRST active-set attempt, give-ups to the work list (k_team_as: the first launch of the default path)
// MODE 1: the whole QP for every instance of the batch - interior-point iterations, with the active-set attempts in between
//         when qp_polish is on (k_team_qp: qp_polish = 0, an attempt schedule that does not start with an attempt, NMPC_TEAM_SPLIT=0)
// MODE 2: the same for ONE chunk of the work list, continuing after a failed first attempt (k_team_qp_list: the second launch)
// MODE 3: ONE step of a work-list instance whose factorisation came from the block-parallel launches of nmpc_block.hpp (k_team_tail:
//         long horizons) - no factor sweep in here
template <bool SHARED, bool TRAJ, bool LDSC, class TI, int MODE = 0>
__device__ __forceinline__ bool team_as(const Consts<double> &c, const Work<double> &w, const Inputs<TI> &in,
                                        const Outputs<TI> &out, const TeamWork<double> &tw, const WorkList &wl,
                                        int B, int tpw, double *smem, int lds_stride, int lstg, int lm_off, int inst_ov = -2,
                                        const TailCtx tcx = TailCtx())
{
    // lds_stride: doubles of LDS per team (carve below + the stage cache at lm_off: [lstg][LMR]); lstg: the factors of stages 0 .. lstg-1 -
    // written last by the backward sweep and read first by the forward sweep - stay in LDS and never reach HBM
    // SHARED: cold start with one linearisation for all stages (x_k = x0, u_k = 0 folded at compile time)
    // TRAJ:   the caller wants x_out / u_out: the forward sweep also leaves xhat_k and every candidate input
    // TI:     element type of the caller's arrays: double, or float (NMPC_DTYPE_F32IO: the arithmetic stays FP64)
    // LDSC:   the build carries the LDS stage cache (one wave per SIMD: 40 KB of LDS per wave); without it
    //         lstg must be 0 and the code is the plain HBM-scratch form (two waves per SIMD: 20 KB per wave)
    using T = double;
    using NoPins = std::integral_constant<bool, false>;
    using WithPins = std::integral_constant<bool, true>;
    constexpr int LMR = MODE == 0 ? AS_LM_ROWS : IP_LM_ROWS;
    const int LDS_T = lds_stride;
    NMPC_PROF_BEGIN
    const int tid = threadIdx.x, team = (tid >> 2) & 3, r = ((tid >> 4) << 2) | (tid & 3);   // the MFMA block layout
    const int ta = r >> 2, tc = r & 3, j = tc;
    const int rr = r < NX ? r : NX - 1;
    const bool rowl = r < NX, cmpl = r < NU;
    int inst = MODE >= 2 ? inst_ov : blockIdx.x * tpw + team;
    const bool valid = MODE >= 2 ? (inst >= 0 && inst < B) : (team < tpw && inst < B);
    if (!valid) inst = B - 1;             // idle teams read the inputs of the last instance ...
    const int winst = valid ? inst : w.Bp;   // ... and work in a spare workspace row, so that no store of a sweep is predicated
    const int lane = inst;                // profiling slot (NMPC_PROFILE builds)
    (void)lane;
    const int N = c.N;
    T *S = smem + team * LDS_T;
    T *sAd = S + A_AD, *sB = S + A_B, *sbv = S + A_BV, *sHg = S + A_HG, *sh = S + A_H, *sXh = S + A_XH;
    T *sRed = S + A_RED, *sEv = S + A_EV, *sLM = S + lm_off;
    const bool warm = !SHARED && in.x_init != nullptr && in.u_init != nullptr;
    const TI *x0p = in.x0 + (size_t)inst * NX;
    const TI *yr = in.yref_bcast ? in.yref : in.yref + (size_t)inst * N * NY;
    const TI *ye = in.yref_bcast ? in.yref_e : in.yref_e + (size_t)inst * NX;
    const TI *xi = warm ? in.x_init + (size_t)inst * (N + 1) * NX : x0p;
    const TI *ui = warm ? in.u_init + (size_t)inst * N * NU : x0p;
    T *const tLM_own = tw.tLM + (size_t)winst * N * TLM_ROWS, *const tIV_own = tw.tIV + (size_t)winst * N * IV_ROWS;
    const int ckpt = c.polish_ckpt;
    T *const tP_own = tw.tP ? tw.tP + (size_t)winst * (ckpt + 1) * TP_ROWS : nullptr;
    T *tAB = SHARED ? nullptr : w.tAB + (size_t)winst * N * TAB_ROWS;
    // a team that has finished keeps sweeping with its wave (the MFMAs are wave-wide) but must not touch its
    // accepted results: from then on it works in the spare row too (pointers switched once per pass)
    T *const tLM_spare = tw.tLM + (size_t)w.Bp * N * TLM_ROWS, *const tIV_spare = tw.tIV + (size_t)w.Bp * N * IV_ROWS;
    T *const tP_spare = tw.tP ? tw.tP + (size_t)w.Bp * (ckpt + 1) * TP_ROWS : nullptr;
    const bool have_tP = tw.tP != nullptr;      // (a kernel argument: branches on it are scalar; on the per-lane pointer tP they were divergent)
    T *tLM = tLM_own, *tIV = tIV_own, *tP = tP_own;

    int natR[4], natC[4];
    NMPC_UNROLL for (int t = 0; t < 4; t++) { natR[t] = nat_of(t, ta); natC[t] = nat_of(t, tc); }
    // ---- per-lane constants, taken into vector registers once
    T dt_v = c.dt, kkt_v = c.kkt_tol;
    asm volatile("" : "+v"(dt_v), "+v"(kkt_v));
    const T x0r = (T)x0p[rr];
    // (c lives in device memory: an entry picked by a lane-dependent index is ONE vector load, where the by-value
    // constant block cost a 13-way select chain per entry and 190 scalar registers - 584 of the 1413 instructions
    // in front of the first MFMA were scalar-register spill traffic)
    const T Wq_r = c.Wq[rr], WqN_r = c.WqN[rr];
    const T Wr_a = c.Wr[ta];
    const T lbj = c.lbu[j], ubj = c.ubu[j];
    const T lb_a = c.lbu[ta], ub_a = c.ubu[ta], Rd_a = c.Rd[ta];
    T Qdg[4];                              // diagonal of the stage Hessian in tile layout
    NMPC_UNROLL for (int t = 0; t < 4; t++) {
        const T qd = c.Qd[natR[t] >= 0 ? natR[t] : 0];
        Qdg[t] = (ta == tc && natR[t] >= 0) ? qd : T(0);
    }
    const int polish_passes = c.polish_passes;
    // where lane (a,c) finds its entries of column / row 15 of the stage cost in the gradient buffer sh: a natural row,
    // or the slot that holds 0.0 (one LDS read and no select per entry in every factor stage)
    constexpr int ZSLOT = A_RED + 32 - A_H;
    int iq_col[4], iq_row[4];
    NMPC_UNROLL for (int t = 0; t < 4; t++) {
        iq_col[t] = (tc == 3 && natR[t] >= 0) ? natR[t] : ZSLOT;
        iq_row[t] = (ta == 3 && natC[t] >= 0) ? natC[t] : ZSLOT;
    }
    if (r == 0) sRed[32] = T(0);
    const T Idt = (ta == tc) ? T(1) : T(0);
    StageLane SL;                                     // per-lane constants of the factor stage (nmpc_stage.hpp)
    SL.ta = ta; SL.tc = tc; SL.dt_v = dt_v; SL.Idt = Idt; SL.Ihalf = (ta == tc) ? T(0.5) : T(0); SL.Rd_a = Rd_a; SL.lb_a = lb_a; SL.ub_a = ub_a; SL.lbj = lbj; SL.ubj = ubj;
    NMPC_UNROLL for (int t = 0; t < 4; t++) { SL.natR[t] = natR[t]; SL.Qdg[t] = Qdg[t]; SL.iq_col[t] = iq_col[t]; SL.iq_row[t] = iq_row[t]; }
    NMPC_STAMP(2)

    // =========================== preparation: linearise Ns shooting intervals
    auto prepare = [&](auto ms_tag) {
        using EV = EvLayout<decltype(ms_tag)::value>;
        constexpr int AS_EV = EV::STRIDE, EV_T2 = EV::T2, EV_B = EV::Bk, EV_RE = EV::RE, EV_ZERO = EV::ZERO, AS_CH = EV::CHUNK;
        const int Ns = SHARED ? 1 : N;
        JacCoef<T> jk;
        jac_coef(c, r, jk);
        const T hh = T(0.5) * c.h, hstep = c.h, inv_mass = c.inv_mass;
        const int nsteps = c.steps;
        constexpr int CH = SHARED ? 1 : AS_CH;
        for (int k0 = 0; k0 < Ns; k0 += CH) {
            // ---- phase A: lane r < CH integrates the state of stage k0 + r and leaves the evaluation points
            {
                const int k = k0 + r;
                const bool mine = r < CH && k < Ns;
                const int kk = mine ? k : 0;
                T xs[NX], us[NU], xn1[NX];
                NMPC_UNROLL for (int i = 0; i < NX; i++) xs[i] = (T)((warm && kk > 0) ? xi[(size_t)kk * NX + i] : x0p[i]);   // stage 0 is pinned to x0
                NMPC_UNROLL for (int i = 0; i < NU; i++) us[i] = warm ? (T)ui[(size_t)kk * NU + i] : T(0);
                NMPC_UNROLL for (int i = 0; i < NX; i++) xn1[i] = (T)(warm ? xi[(size_t)(kk + 1) * NX + i] : x0p[i]);
                T *ev = sEv + (SHARED ? 0 : (r < CH ? r : 0) * AS_EV);
                for (int st = 0; st < nsteps; st++) {
                    T f1[NX], xm[NX], f2[NX];
                    model_f(c, xs, us, f1);
                    NMPC_UNROLL for (int i = 0; i < NX; i++) xm[i] = xs[i] + hh * f1[i];
                    if (mine) {
                        NMPC_UNROLL for (int i = 0; i < 7; i++) { ev[st * 14 + i] = xs[6 + i]; ev[st * 14 + 7 + i] = xm[6 + i]; }
                        thrust_dir(inv_mass, xs, ev + EV_RE + st * 6);
                        thrust_dir(inv_mass, xm, ev + EV_RE + st * 6 + 3);
                    }
                    model_f(c, xm, us, f2);
                    NMPC_UNROLL for (int i = 0; i < NX; i++) xs[i] += hstep * f2[i];
                }
                if (mine) {
                    ev[EV_T2] = T(2) * (us[0] + us[1] + us[2] + us[3]) * inv_mass; ev[EV_ZERO] = T(0);
                    if (SHARED) {
                        NMPC_UNROLL for (int i = 0; i < NX; i++) sbv[i] = xs[i] - xn1[i];
                    } else {
                        NMPC_UNROLL for (int i = 0; i < NX; i++) ev[EV_B + i] = xs[i] - xn1[i];
                    }
                }
            }
            NMPC_WSYNC();
            NMPC_STAMP(3)
            // ---- phase B: forward sensitivities of the stages of the chunk, tile form, TWO stages at a time: their MFMA
            // chains are independent, so one stage's variational-equation products run in the latency of the other's
            constexpr int PB = SHARED ? 1 : 2;
            const int ire = ta < 3 ? EV_RE + ta : EV_ZERO, sre = ta < 3 ? 3 : 0;     // lane's entry of R e3 / m at point p: ire + sre p (row 3: the zero slot)
            for (int e0 = 0; e0 < CH; e0 += PB) {
                if (k0 + e0 >= Ns) break;
                const T *ev[PB];
                T t2[PB], Sx[PB][4][3];
                NMPC_UNROLL for (int q = 0; q < PB; q++) {
                    const int e = (e0 + q < CH && k0 + e0 + q < Ns) ? e0 + q : e0;      // an odd tail repeats the last stage
                    ev[q] = sEv + e * AS_EV;
                    t2[q] = ev[q][EV_T2];
                    NMPC_UNROLL for (int rt = 0; rt < 4; rt++) {
                        NMPC_UNROLL for (int ct = 0; ct < 3; ct++) Sx[q][rt][ct] = 0;
                    }
                    Sx[q][2][0] = (ta == tc) ? T(1) : T(0);                  // d q / d q = I
                    Sx[q][3][1] = (ta == tc && ta < 3) ? T(1) : T(0);        // d w / d w = I
                }
                for (int st = 0; st < nsteps; st++) {
                    T K[PB][4][3], Sm[PB][4][3];
                    NMPC_UNROLL for (int q = 0; q < PB; q++) {
                        T a1, a2, a3, a4, a5;
                        jac_tiles_ev(jk, ev[q] + st * 14, ev[q][ire + sre * (2 * st)], t2[q], a1, a2, a3, a4, a5);
                        vde_tiles<T, true>(a1, a2, a3, a4, a5, jk.fu, Sx[q], K[q]);
                    }
                    NMPC_UNROLL for (int q = 0; q < PB; q++) {
                        NMPC_UNROLL for (int rt = 0; rt < 4; rt++) {
                            NMPC_UNROLL for (int ct = 0; ct < 3; ct++) Sm[q][rt][ct] = (rt == 3 && ct == 0) ? T(0) : Sx[q][rt][ct] + hh * K[q][rt][ct];
                        }
                    }
                    NMPC_UNROLL for (int q = 0; q < PB; q++) {
                        T a1, a2, a3, a4, a5;
                        jac_tiles_ev(jk, ev[q] + st * 14 + 7, ev[q][ire + sre * (2 * st + 1)], t2[q], a1, a2, a3, a4, a5);
                        vde_tiles<T, true>(a1, a2, a3, a4, a5, jk.fu, Sm[q], K[q]);
                    }
                    NMPC_UNROLL for (int q = 0; q < PB; q++) {
                        NMPC_UNROLL for (int rt = 0; rt < 4; rt++) {
                            NMPC_UNROLL for (int ct = 0; ct < 3; ct++) { if (!(rt == 3 && ct == 0)) Sx[q][rt][ct] += hstep * K[q][rt][ct]; }    // (3,0): d omega / d q = 0
                        }
                    }
                }
                NMPC_UNROLL for (int q = 0; q < PB; q++) {
                    const int k = k0 + e0 + q;
                    if (q == 0 || (e0 + q < CH && k < Ns)) {
                        if (SHARED) {           // natural layout in LDS: Ad rows [13][8] | B rows [13][4] (b came from phase A)
                            NMPC_UNROLL for (int rt = 0; rt < 4; rt++) {
                                if (natR[rt] >= 0) {
                                    S[A_AD + natR[rt] * 8 + tc] = Sx[q][rt][0];
                                    S[A_AD + natR[rt] * 8 + 4 + tc] = tc < 3 ? Sx[q][rt][1] : T(0);
                                    S[A_B + natR[rt] * NU + tc] = Sx[q][rt][2];
                                }
                            }
                            if (MODE == 3) {
                                // the block-parallel factorisation reads the stage as tiles from the HBM scratch: stage 0 of the instance's rows
                                T *a = w.tAB + (size_t)winst * N * TAB_ROWS + r;
                                NMPC_UNROLL for (int rt = 0; rt < 4; rt++) {
                                    const T bq = sbv[natR[rt] >= 0 ? natR[rt] : 0];
                                    const T c15 = natR[rt] >= 0 ? bq : ((rt == 3 && ta == 3) ? T(1) : T(0));
                                    if (rt < 3) a[(rt * 3 + 0) * 16] = Sx[q][rt][0];
                                    a[(rt * 3 + 1) * 16] = tc < 3 ? Sx[q][rt][1] : c15;
                                    a[(rt * 3 + 2) * 16] = Sx[q][rt][2];
                                }
                            }
                        } else {
                            // per-stage linearisation: the 12 tiles of [Aq | Aw b | B] of the padded homogeneous form go to
                            // the HBM scratch as tiles - tile (rt,ct) at [(3 rt + ct) 16 + r]: the factor sweep reads its
                            // operand tiles back with one load each, the forward sweep reads them transposed by
                            // swapping the lane index.  (Pad rows of the sensitivities are zero by construction; b_k
                            // and the homogeneous 1 sit in column 15.)
                            T *a = tAB + (size_t)k * TAB_ROWS + r;
                            NMPC_UNROLL for (int rt = 0; rt < 4; rt++) {
                                const T bq = ev[q][EV_B + (natR[rt] >= 0 ? natR[rt] : 0)];
                                const T c15 = natR[rt] >= 0 ? bq : ((rt == 3 && ta == 3) ? T(1) : T(0));
                                if (rt < 3) a[(rt * 3 + 0) * 16] = Sx[q][rt][0];          // (tile (3,0) is zero and never read)
                                a[(rt * 3 + 1) * 16] = tc < 3 ? Sx[q][rt][1] : c15;
                                a[(rt * 3 + 2) * 16] = Sx[q][rt][2];
                            }
                        }
                    }
                }
            }
            NMPC_WSYNC();
            NMPC_STAMP(4)
        }
    };
    if (SHARED || MODE != 3) {       // (MODE 3, per-stage variant: the tiles the first launch left in the HBM scratch are current)
        if (c.steps <= 2) prepare(std::integral_constant<int, 2>{}); else prepare(std::integral_constant<int, AS_MAX_STEPS>{});
    }
    __syncthreads();          // stage matrices (LDS, or this wave's own global rows) visible to every lane
    NMPC_STAMP(7)

    // stage tiles of the per-stage variant: HBM scratch -> registers, one stage ahead (lr = r: as stored; lr = 4c + a:
    // every tile transposed)
    T pfs[12];
    auto fetch_stage = [&](int k, int lr) {
        const T *a = tAB + (size_t)k * TAB_ROWS + lr;
        NMPC_UNROLL for (int t = 0; t < 12; t++) pfs[t] = a[t * 16];
    };
    const int rT = tc * 4 + ta;
    // ... and its four input tiles once more, transposed (tile (kt, 2) through the swapped lane index): the operand of b += B v in the factor
    // stages that have pinned inputs or barrier terms
    T pft[4];
    auto fetch_bt = [&](int k) {
        const T *a = tAB + (size_t)k * TAB_ROWS + rT;
        NMPC_UNROLL for (int kt = 0; kt < 4; kt++) pft[kt] = a[(kt * 3 + 2) * 16];
    };
    // linearisation point of stage k, natural row rr / input comp (cold start: x_k = x0, u_k = 0)
    auto xlin = [&](int k) -> T { return (warm && k > 0) ? (T)xi[(size_t)k * NX + rr] : x0r; };
    auto ulin = [&](int k, int comp) -> T { return warm ? (T)ui[(size_t)k * NU + comp] : T(0); };

    int status = 0, npol = 0, pass_in_attempt = 0;
    int k_top = N - 1;       // highest stage this team's next backward sweep has to refactorise
    int ck_valid = 0;        // checkpoints 1..ck_valid of this team are current
    // per-team mode: active-set pass | finished | (MODE 0) handed to the work list | interior-point iteration | waiting for the
    // wave's next active-set phase
    enum { M_POL = 1, M_DONE = 2, M_GIVEUP = 3, M_IPM = 4, M_WAIT = 5 };
    int mode = valid ? M_POL : M_DONE;
    int pass = 0;            // wave-uniform pass counter: all live teams of a wave are in the same pass
    bool nopins_pass = true; // wave-uniform: the pass starts from "all inputs free" (first pass of a first attempt)
    T u0_cand = 0;           // lane (a,0): candidate command of input a from the latest forward sweep
    // lanes of this team in a wave-wide ballot
    const unsigned long long team_mask = 0x000F000F000F000Full << (4 * team);
    // state of the sweep in flight (set by the pass / iteration that runs it)
    bool pol = false, pol2 = false, ok = true, nanp = false, viol = false, heavy = false;
    int ks = N - 1, wnd = 0, kchgB = -1;
    T xh = 0;
    // growth certificate (nmpc_config.qp_growth_max): gm = max |B'PB| of the sweep in flight as this lane sees it, gbase = the
    // team-wide value of the FIRST factorisation of the solve (0 = none yet)
    T gm = 0, gbase = 0;
    bool tripped = false;    // the certificate ended an attempt of this team: no further attempt is made
    bool from_ua = MODE == 0; // the result is an accepted active-set solution (candidate inputs, xhat of its forward sweep)
    // interior-point iteration (IPMK kernels): scalars of the corrector
    T sigmu = 0;
    using NoIpm = std::integral_constant<bool, false>;
    using Ipm = std::integral_constant<bool, true>;
    using NoWarm = std::integral_constant<bool, false>;
    using Warm = std::integral_constant<bool, true>;
    // warm start of the interior point from an attempt that ran out of passes (nmpc_config.qp_warm_start; oracle ORC_WARM_*)
    constexpr double WARM_DELTA = 1e-3, WARM_MU = 1e-3;
    const T wd_a = T(WARM_DELTA) * (ub_a - lb_a);
    bool anyp = false;       // the pass in flight had pinned inputs (lane-local; reduced per team)
    bool warm_avail = false; // slots 24..35 of this team hold a warm start the interior point has not taken yet
    bool warm_derived = false; // the interior point's iterate descends from a warm start, not from the cold point
    const T iw_a = T(1) / (ub_a - lb_a);          // step sizes are measured against the box width

    // ================= sweep A: backward factorisation in tile form.
    // The shape of the code: a stage is
    // ONE basic block (no predicated store: finished and idle teams work in a spare workspace row), P B and
    // B'P B come first so that the 4x4 Cholesky - a serial chain of ~90
    // vector instructions - runs beside the ~60 MFMAs that do not depend on it, and the first pass (nothing
    // pinned, by construction) is compiled without any of the pin handling.
    auto sweepA = [&](auto pins_tag, auto ipm_tag) {
        constexpr bool PINS = decltype(pins_tag)::value;
        constexpr bool IPMV = decltype(ipm_tag)::value;     // interior-point iteration: barrier terms on the input Hessian, nothing pinned
        T Aq0[4], Aq1b[4], Bt[4], BtT[4];       // BtT: the input tiles transposed (operand of b += B v: pinned values, the interior point's inputs)
        constexpr bool NEED_BT = PINS || IPMV;
        NMPC_UNROLL for (int kt = 0; kt < 4; kt++) BtT[kt] = 0;
        auto load_tiles = [&]() {
            NMPC_UNROLL for (int kt = 0; kt < 4; kt++) {
                const int l = natR[kt] >= 0 ? natR[kt] : 0;
                const bool real = natR[kt] >= 0;
                const T a0 = sAd[l * 8 + tc], a1 = sAd[l * 8 + 4 + (tc < 3 ? tc : 0)], bb = sB[l * 4 + tc], bv_ = sbv[l];
                Aq0[kt] = real ? a0 : T(0);
                Aq1b[kt] = real ? (tc < 3 ? a1 : bv_) : ((kt == 3 && ta == 3 && tc == 3) ? T(1) : T(0));
                Bt[kt] = real ? bb : T(0);
                if (NEED_BT) {
                    const int lc = natC[kt] >= 0 ? natC[kt] : 0;
                    const T bt = sB[lc * 4 + ta];
                    BtT[kt] = natC[kt] >= 0 ? bt : T(0);
                }
            }
        };
        if (SHARED) load_tiles(); else { fetch_stage(ks, r); if (NEED_BT) fetch_bt(ks); }
        T Pt[4][4];
        if (ks == N - 1) {
            // terminal cost: QdN on the diagonal, q_N = WqN (x_N - yref_e) in row / column 15
            sh[r] = WqN_r * (xlin(N) - (T)ye[rr]);
            NMPC_WSYNC();
            NMPC_UNROLL for (int it = 0; it < 4; it++) {
                NMPC_UNROLL for (int jt = 0; jt < 4; jt++) {
                    T v = (it == jt && ta == tc && natR[it] >= 0) ? c.QdN[natR[it] >= 0 ? natR[it] : 0] : T(0);
                    const T qa = sh[natR[it] >= 0 ? natR[it] : 0], qb = sh[natC[jt] >= 0 ? natC[jt] : 0];
                    if (jt == 3 && tc == 3 && natR[it] >= 0) v = qa;
                    if (it == 3 && ta == 3 && natC[jt] >= 0) v = qb;
                    Pt[it][jt] = v;
                }
            }
            NMPC_WSYNC();
        } else {                  // resume from the checkpoint an earlier pass left
            const T *cp = tP + (size_t)(ks + 1) * TP_ROWS + r;
            NMPC_UNROLL for (int it = 0; it < 4; it++) {
                NMPC_UNROLL for (int jt = 0; jt < 4; jt++) Pt[it][jt] = cp[(it * 4 + jt) * 16];
            }
        }
        // scalars of a stage are fetched one stage ahead (global loads stay in flight over the stage):
        // reference row rr, reference / linearisation input of component a; pins variant: the pin codes of inputs a AND c (and the
        // linearisation input of component c) - lane (a,c) then forms the masks, the pinned values and the modified gradient of
        // BOTH its inputs in registers (in round 2 lanes 0..3 computed them and the team exchanged them through LDS: four
        // predicated stores, five reads and a hand-off at the head of every stage)
        const int cu = ta;
        T n_yx = (T)yr[(size_t)ks * NY + rr], n_yu = (T)yr[(size_t)ks * NY + NX + cu];
        T n_xl = xlin(ks), n_ul = ulin(ks, cu);
        T n_pc = PINS ? tIV[ks * IV_ROWS + 16 + j] : T(0);
        T n_pca = PINS ? tIV[ks * IV_ROWS + 16 + ta] : T(0), n_ulc = PINS ? ulin(ks, j) : T(0);
        T n_u = 0, n_ll = 0, n_lu = 0, n_tl = 0, n_tu = 0;                 // the iterate of input a (interior-point variant)
        if (IPMV) {
            const T *ivn = tIV + ks * IV_ROWS;
            n_u = ivn[ta]; { const D2 p_ = ld2(ivn + IVP_L + 2 * ta); n_ll = p_.x; n_lu = p_.y; } { const D2 p_ = ld2(ivn + IVP_T + 2 * ta); n_tl = p_.x; n_tu = p_.y; }
        }
        auto stage = [&](int k, auto last_tag, auto lds_tag) {
            constexpr bool LAST = decltype(last_tag)::value;      // stage 0: no Riccati update needed
            constexpr bool LDSST = decltype(lds_tag)::value;      // the factors of this stage stay in LDS
            if (!SHARED) {
                NMPC_UNROLL for (int kt = 0; kt < 4; kt++) { Aq0[kt] = pfs[kt * 3]; Aq1b[kt] = pfs[kt * 3 + 1]; Bt[kt] = pfs[kt * 3 + 2]; }
                if (NEED_BT) { NMPC_UNROLL for (int kt = 0; kt < 4; kt++) BtT[kt] = pft[kt]; }
                if (!LAST) { fetch_stage(k - 1, r); if (NEED_BT) fetch_bt(k - 1); }
            }
            T *lmk = tLM + k * TLM_ROWS;
            const T ul = n_ul, pc = n_pc, pca = n_pca, ulc = n_ulc, u_it = n_u, ll_it = n_ll, lu_it = n_lu, tl_it = n_tl, tu_it = n_tu;
            // r_k must be a ROUNDED product in both variants (the pins variant passes it through LDS): left to
            // -ffp-contract the first-pass variant fuses it into gu = B'h + r_k, one rounding less, and a result
            // would depend on which variant last factorised a stage - i.e. on the wave-mates of an instance
            // (found by the permutation test at N = 600)
            T rk = Wr_a * (ul - n_yu);
            const T q_r = Wq_r * (n_xl - n_yx);
            asm volatile("" : "+v"(rk));             // (q_r passes through LDS, which rounds it in both variants)
            if (!LAST) {
                n_yx = (T)yr[(size_t)(k - 1) * NY + rr]; n_yu = (T)yr[(size_t)(k - 1) * NY + NX + cu];
                n_xl = xlin(k - 1); n_ul = ulin(k - 1, cu);
                if (PINS) { n_pc = tIV[(k - 1) * IV_ROWS + 16 + j]; n_pca = tIV[(k - 1) * IV_ROWS + 16 + ta]; n_ulc = ulin(k - 1, j); }
                if (IPMV) {
                    const T *ivn = tIV + (k - 1) * IV_ROWS;
                    n_u = ivn[ta]; { const D2 p_ = ld2(ivn + IVP_L + 2 * ta); n_ll = p_.x; n_lu = p_.y; } { const D2 p_ = ld2(ivn + IVP_T + 2 * ta); n_tl = p_.x; n_tu = p_.y; }
                }
            }
            StageIn sin;
            sin.rk = rk; sin.q_r = q_r; sin.ul = ul; sin.ulc = ulc; sin.pc = pc; sin.pca = pca; sin.u_it = u_it; sin.ll_it = ll_it; sin.lu_it = lu_it;
            sin.tl_it = tl_it; sin.tu_it = tu_it;
            StageOut so;
            // the stage itself: nmpc_stage.hpp (one source for this sweep and the block sweeps of nmpc_block.hpp).  Stores happen where they
            // always did: gradient rows of pinned stages while X is formed, the factors as soon as M is final
            auto sink = stage_sink(
                [&](int jt, T a) { lmk[TLM_G + jt * 16 + tc * 4 + ta] = a; },
                [&](T hr) { lmk[TLM_G + 64 + tc * 4 + ta] = hr; },
                [&](const StageOut &f) {
                    NMPC_UNROLL for (int jt = 0; jt < 4; jt++) {
                        // stored where the forward sweep reads it transposed
                        if (LDSST) sLM[k * LMR + jt * 16 + tc * 4 + ta] = f.M[jt]; else lmk[TLM_MT + jt * 16 + tc * 4 + ta] = f.M[jt];
                    }
                    if (LDSST) sLM[k * LMR + 64 + r] = f.Zt; else lmk[TLM_Z + r] = f.Zt;
                    if (IPMV) {              // 1 / d_a for the corrector's solves (lanes (a, c != 0) store the same value to a spare slot)
                        const int rs = ta + (tc == 0 ? 0 : 4);
                        if (LDSST) sLM[k * LMR + 80 + rs] = f.ra; else lmk[TLM_RINV + rs] = f.ra;
                    }
                });
            riccati_factor_stage<PINS, IPMV, LAST, true>(SL, sh, sHg, r, Aq0, Aq1b, Bt, BtT, sin, pol, true, Pt, gm, ok, nanp, so, sink);
            if (!LAST) {
                if (!IPMV && have_tP && k <= wnd) {
                    T *cp = tP + (size_t)k * TP_ROWS + r;
                    NMPC_UNROLL for (int it = 0; it < 4; it++) {
                        NMPC_UNROLL for (int jt = 0; jt < 4; jt++) cp[(it * 4 + jt) * 16] = Pt[it][jt];
                    }
                }
            }
            NMPC_WSYNC();
        };
        using Fl = std::integral_constant<bool, false>;
        using Tr = std::integral_constant<bool, true>;
        int k = ks;
        if constexpr (LDSC) {
            const int kg = lstg > 1 ? lstg : 1;
            for (; k >= kg; k--) stage(k, Fl{}, Fl{});        // factors to HBM
            for (; k > 0; k--) stage(k, Fl{}, Tr{});          // factors stay in LDS
            if (lstg > 0) stage(0, Tr{}, Tr{}); else stage(0, Tr{}, Fl{});
        } else {
            for (; k > 0; k--) stage(k, Fl{}, Fl{});
            stage(0, Tr{}, Fl{});
        }
    };
    // range and start state of the forward sweep (MODE 3, block-parallel forward: one team sweeps one block; otherwise the horizon)
    int fwd_s = 0, fwd_e = N;
    const T *fwd_x = nullptr;
    // ================= sweep B: forward solve + KKT check in tile form: a stage
    // is one basic block; operands arrive two stages ahead in two alternating register sets
    auto sweepB = [&](auto pins_tag, auto ipm_tag, auto warm_tag) {
        constexpr bool PINS = decltype(pins_tag)::value;
        constexpr bool IPMV = decltype(ipm_tag)::value;     // predictor of an interior-point iteration: affine target, step length terms
        constexpr bool WARM = decltype(warm_tag)::value;    // last pass of an attempt: also leaves the interior point's warm start (slots 24..35)
        T rmaxB = T(1), s2B = 0;                            // largest inverse step length (floor 1), complementarity of the affine step
        T AT2[4], AT3[4], BT[4];
        auto load_tiles_T = [&]() {
            NMPC_UNROLL for (int it = 0; it < 4; it++) {
                const int l = natC[it] >= 0 ? natC[it] : 0;
                const bool real = natC[it] >= 0;
                const T a2 = sAd[l * 8 + ta], a3 = sAd[l * 8 + 4 + (ta < 3 ? ta : 0)], bb = sB[l * 4 + ta], bv_ = sbv[l];
                AT2[it] = real ? a2 : T(0);
                AT3[it] = real ? (ta < 3 ? a3 : bv_) : ((it == 3 && tc == 3 && ta == 3) ? T(1) : T(0));
                BT[it] = real ? bb : T(0);
            }
        };
        if (SHARED) load_tiles_T();
        T xt[4];
        NMPC_UNROLL for (int t = 0; t < 4; t++) xt[t] = (t == 3 && ta == 3 && tc == 0) ? T(1) : T(0);
        if (MODE == 3 && fwd_x) {             // block-parallel forward sweep: the state at the start of this team's block (lanes (a,0))
            NMPC_UNROLL for (int t = 0; t < 4; t++) xt[t] = tc == 0 ? fwd_x[t * 4 + ta] : T(0);
        }
        // lanes (a, c != 0) carry no part of xbar / u: their stores go to spare slots (xhat pad slot 13, tIV slot 0)
        int xslot[4];
        NMPC_UNROLL for (int t = 0; t < 4; t++) xslot[t] = (tc == 0 && natR[t] >= 0) ? natR[t] : 13;
        const int cslot = tc == 0 ? 12 + ta : 20 + ta, pslot = tc == 0 ? 16 + ta : 20 + ta;
        struct Ops { T mt[4], z, ul, pc, u, ll, lu, tl, tu, ab[SHARED ? 1 : 12]; };
        const int kl = LDSC ? (lstg < N ? lstg : N) : 0;     // factors of stages [0, kl) come from LDS, [kl, N) from HBM
        // operands of stage kq from the HBM scratch (clamped index: prefetches run past the horizon).  Per-stage
        // linearisation: the stage tiles (transposed) travel with them - two to four stages ahead, where one stage
        // ahead left the 0.33 us stage waiting on a 1 us load - and the loop below covers the LDS-cached stages too
        auto fetch_ops = [&](int kq, Ops &o) {
            const int k = kq < N ? kq : N - 1;   // clamped, NOT skipped: a branch around the loads makes the compiler's
                                                 // wait-count model assume the fewest loads in flight, and every stage
                                                 // then waits for the prefetch issued just before it
            const T *lmn = tLM + k * TLM_ROWS;
            if (SHARED || k >= kl) {
                NMPC_UNROLL for (int jt = 0; jt < 4; jt++) o.mt[jt] = lmn[TLM_MT + jt * 16 + r];   // Mbar[c][4jt+a]
                o.z = lmn[TLM_Z + r];                                                            // (L^-1)[a][c]
            }
            o.ul = ulin(k, ta);
            o.pc = PINS ? tIV[k * IV_ROWS + 16 + ta] : T(0);
            if (IPMV) { const T *ivn = tIV + k * IV_ROWS; o.u = ivn[ta]; { const D2 p_ = ld2(ivn + IVP_L + 2 * ta); o.ll = p_.x; o.lu = p_.y; } { const D2 p_ = ld2(ivn + IVP_T + 2 * ta); o.tl = p_.x; o.tu = p_.y; } }
            if (!SHARED) {
                const T *a = tAB + (size_t)k * TAB_ROWS + rT;
                NMPC_UNROLL for (int t = 0; t < 12; t++) o.ab[t] = a[t * 16];
            }
        };
        // ... and from the LDS stage cache (scalars still come from global memory, one stage ahead)
        auto fetch_ops_lds = [&](int k, Ops &o) {
            const T *lmn = sLM + k * LMR;
            NMPC_UNROLL for (int jt = 0; jt < 4; jt++) o.mt[jt] = lmn[jt * 16 + r];
            o.z = lmn[64 + r];
        };
        auto stageB = [&](int k, Ops &o) {
            if (!SHARED) {
                NMPC_UNROLL for (int it = 0; it < 4; it++) { AT2[it] = o.ab[it * 3]; AT3[it] = o.ab[it * 3 + 1]; BT[it] = o.ab[it * 3 + 2]; }
                if (LDSC) { if (k < kl) fetch_ops_lds(k, o); }
            }
            T *ivk = tIV + k * IV_ROWS;
            const T ul = o.ul, pc = o.pc;
            if (TRAJ && !IPMV) {           // xhat_k for the output sweep
                T *xs = tLM + k * TLM_ROWS + 66;
                NMPC_UNROLL for (int t = 0; t < 4; t++) xs[xslot[t]] = xt[t];
            }
            T xn[4];
            xn[0] = xt[0] + dt_v * xt[1]; xn[1] = xt[1]; xn[2] = 0; xn[3] = 0;
            NMPC_UNROLL for (int it = 0; it < 4; it++) xn[it] = mfma44(AT3[it], xt[3], it < 3 ? mfma44(AT2[it], xt[2], xn[it]) : xn[it]);   // tile (3,0) of Abar is zero
            const T v = mfma44(o.mt[2], xt[2], mfma44(o.mt[0], xt[0], T(0)))
                      + mfma44(o.mt[3], xt[3], mfma44(o.mt[1], xt[1], T(0)));
            const T ut = mfma44_na(o.z, v, T(0));                         // lane (a,0): u_a = -(L^-T v)_a
            if constexpr (IPMV) {
                // the sweep solved for the STEP of the inputs: the state moves with the iterate's input plus the step (lanes (a,0))
                const T uf = ut + (tc == 0 ? o.u : T(0));
                NMPC_UNROLL for (int it = 0; it < 4; it++) xn[it] = mfma44(BT[it], uf, xn[it]);
                // affine-scaling step of input a in lane (a,0): step-length terms of the predictor, step kept for the corrector
                const T d = ut;                                             // (the slacks move with it: dt_l = d, dt_u = -d)
                const Pair<T> pr(o.ll, o.lu, o.tl, o.tu);
                const T dla = -o.ll - pr.kl * d, dua = -o.lu + pr.ku * d;
                // inverse step lengths: -d/tl, d/tu, -dla/ll = 1 + d/tl, -dua/lu = 1 - d/tu
                const T a1 = d * pr.itl, a2 = d * pr.itu;
                rmaxB = fmax(rmaxB, fmax(fmax(-a1, a2), fmax(T(1) + a1, T(1) - a2)));
                s2B += dla * d - dua * d;
                ivk[cslot] = d;
                (void)pc; (void)ul;
            } else {
                NMPC_UNROLL for (int it = 0; it < 4; it++) xn[it] = mfma44(BT[it], ut, xn[it]);
                // KKT check of the pass, input a in lane (a,0): a free input must sit inside its box; a pinned one
                // must have a multiplier of the right sign (gradient from the rows the factor sweep left)
                const T uj = ut;
                const T lo = lb_a - ul, hi = ub_a - ul;
                const T tolb = kkt_v * (T(1) + fabs(lo) + fabs(hi));
                T npc = uj < lo - tolb ? T(-1) : (uj > hi + tolb ? T(1) : T(0));
                T ue = uj;
                T gpin = 0;                                                     // multiplier estimate of a pinned input (gradient of the QP's Lagrangian)
                bool nanq = !(uj == uj);
                if (PINS) {
                    const bool pin_here = pol2 && pc != T(0);
                    const T vpin = pc < T(0) ? lo : hi;
                    ue = pin_here ? vpin : uj;                                 // pinned inputs sit exactly on the bound
                    if (__ballot(tc == 0 && pin_here) != 0) {
                        const T *lmn = tLM + k * TLM_ROWS;
                        T cG[5];
                        NMPC_UNROLL for (int g5 = 0; g5 < 5; g5++) cG[g5] = lmn[TLM_G + g5 * 16 + r];
                        T rka = Wr_a * (ul - (T)yr[(size_t)k * NY + NX + ta]);
                        asm volatile("" : "+v"(rka));
                        const T uf = (tc == 0 && !pin_here) ? ue : T(0);                    // mask u
                        T g = mfma44(cG[4], uf, T(0));
                        T g2 = mfma44(cG[1], xt[1], mfma44(cG[0], xt[0], T(0)));
                        g = mfma44(cG[3], xt[3], mfma44(cG[2], xt[2], g));
                        g += g2 + Rd_a * ue + rka;
                        const T tolg = kkt_v * (T(1) + fabs(g));
                        const bool wrong = (pc < T(0) && g < -tolg) || (pc > T(0) && g > tolg);
                        npc = pin_here ? (wrong ? T(0) : pc) : npc;
                        nanq |= !(g == g);
                        gpin = pin_here ? g : T(0);
                    }
                    if (WARM) {
                        // warm start of the interior-point iteration in case this attempt ends here without an accepted pass
                        // (oracle ocpqp_polish): a pinned input keeps its multiplier estimate g and takes the slack mu / g that
                        // puts the pair ON the central path of mu = 1e-3 (at most 1e-3 of the box width: a small g is floored
                        // instead); every other pair sits on that path by construction
                        const T gl = pc < T(0) ? gpin : T(0), gh = pc > T(0) ? -gpin : T(0);
                        T tl = wd_a, th = wd_a;
                        tl = gl * tl > T(WARM_MU) ? T(WARM_MU) * fast_rcp(gl > T(0) ? gl : T(1)) : tl;
                        th = gh * th > T(WARM_MU) ? T(WARM_MU) * fast_rcp(gh > T(0) ? gh : T(1)) : th;
                        const T v = fmin(fmax(ue, lo + tl), hi - th);
                        ivk[tc == 0 ? 24 + ta : 20 + ta] = v;
                        ivk[tc == 0 ? 28 + ta : 20 + ta] = T(WARM_MU) * fast_rcp(v - lo);
                        ivk[tc == 0 ? 32 + ta : 20 + ta] = T(WARM_MU) * fast_rcp(hi - v);
                        anyp |= pin_here;
                    }
                }
                if (TRAJ) ivk[cslot] = ue;                                     // every candidate input
                ivk[pslot] = npc;
                if (k == 0) u0_cand = pol2 ? ue : u0_cand;
                heavy |= nanq;
                viol |= npc != pc;
                kchgB = (npc != pc) ? k : kchgB;                               // ascending k: the last one is the highest
            }
            NMPC_UNROLL for (int t = 0; t < 4; t++) xt[t] = xn[t];
        };
        // Stages in flight: two sets of H = 2.  (The stage sits on its dependent MFMA chain - 0.3 us whether the operands
        // come from LDS or from HBM; 2 x 3 and 2 x 4 stages in flight, also issued ahead of the LDS phase, measured the
        // same 61-63 M solves/s.)
        constexpr int H = 2;
        constexpr bool PRE = false;
        Ops oa[H], ob[H];
        const int kb = MODE == 3 ? fwd_s : (SHARED ? kl : 0);   // first stage of the main loop
        const int fe = MODE == 3 ? fwd_e : N;                   // ... and the stage behind its last (MODE 3: this team's block)
        NMPC_UNROLL for (int i = 0; i < H; i++) fetch_ops(kb + i, oa[i]);
        if (PRE) { NMPC_UNROLL for (int i = 0; i < H; i++) fetch_ops(kb + H + i, ob[i]); }
        if constexpr (LDSC && SHARED) {
            if (kl > 0) {
                Ops ol;
                T n_ul = ulin(0, ta), n_pc = PINS ? tIV[16 + ta] : T(0);
                T n_u = 0, n_ll = 0, n_lu = 0, n_tl = 0, n_tu = 0, m_u = 0, m_ll = 0, m_lu = 0, m_tl = 0, m_tu = 0;      // interior-point variant: the iterate two stages ahead
                if (IPMV) {
                    n_u = tIV[ta]; { const D2 p_ = ld2(tIV + IVP_L + 2 * ta); n_ll = p_.x; n_lu = p_.y; } { const D2 p_ = ld2(tIV + IVP_T + 2 * ta); n_tl = p_.x; n_tu = p_.y; }
                    const T *iv1 = tIV + (1 < N ? 1 : 0) * IV_ROWS;
                    m_u = iv1[ta]; { const D2 p_ = ld2(iv1 + IVP_L + 2 * ta); m_ll = p_.x; m_lu = p_.y; } { const D2 p_ = ld2(iv1 + IVP_T + 2 * ta); m_tl = p_.x; m_tu = p_.y; }
                }
                // the factors of stage k + 1 leave LDS before stage k computes (two operand sets, alternating): a stage is a short
                // chain of dependent MFMAs and would otherwise open with an exposed LDS round trip
                Ops ol2;
                auto scalars = [&](Ops &o, int k) {
                    o.ul = n_ul; o.pc = n_pc; o.u = n_u; o.ll = n_ll; o.lu = n_lu; o.tl = n_tl; o.tu = n_tu;
                    const int kn = k + 1 < N ? k + 1 : k;
                    n_ul = ulin(kn, ta);
                    if (PINS) n_pc = tIV[kn * IV_ROWS + 16 + ta];
                    if (IPMV) {
                        n_u = m_u; n_ll = m_ll; n_lu = m_lu; n_tl = m_tl; n_tu = m_tu;
                        const T *ivn = tIV + (k + 2 < N ? k + 2 : N - 1) * IV_ROWS;
                        m_u = ivn[ta]; { const D2 p_ = ld2(ivn + IVP_L + 2 * ta); m_ll = p_.x; m_lu = p_.y; } { const D2 p_ = ld2(ivn + IVP_T + 2 * ta); m_tl = p_.x; m_tu = p_.y; }
                    }
                };
                fetch_ops_lds(0, ol);
                for (int k = 0; k < kl; k += 2) {
                    if (k + 1 < kl) fetch_ops_lds(k + 1, ol2);
                    scalars(ol, k);
                    stageB(k, ol);
                    if (k + 1 < kl) {
                        if (k + 2 < kl) fetch_ops_lds(k + 2, ol);
                        scalars(ol2, k + 1);
                        stageB(k + 1, ol2);
                    }
                }
            }
        }
        for (int k0 = kb; k0 < fe; k0 += 2 * H) {
            if (!PRE || k0 != kb) { NMPC_UNROLL for (int i = 0; i < H; i++) fetch_ops(k0 + H + i, ob[i]); }
            NMPC_UNROLL for (int i = 0; i < H; i++) { if (k0 + i < fe) stageB(k0 + i, oa[i]); }
            NMPC_UNROLL for (int i = 0; i < H; i++) fetch_ops(k0 + 2 * H + i, oa[i]);
            NMPC_UNROLL for (int i = 0; i < H; i++) { if (k0 + H + i < fe) stageB(k0 + H + i, ob[i]); }
        }
        // back to one natural row per lane
        if (tc == 0) {
            NMPC_UNROLL for (int t = 0; t < 4; t++)
                if (natR[t] >= 0) sXh[natR[t]] = xt[t];
        }
        NMPC_WSYNC();
        xh = sXh[rr];
        if (tc == 0) sRed[28 + ta] = (T)kchgB;
        if (IPMV) { if (tc == 0) { sRed[4 + ta] = rmaxB; sRed[8 + ta] = s2B; } }
    };
    // ================= interior-point iteration (kernels of MODE 1 / 2 only; generic lambdas: never instantiated in MODE 0).
    // In the shape of the active-set sweeps above: a stage is one basic block, idle
    // and finished teams work in the spare workspace row, factors of the leading stages come from the LDS stage cache.
    // H_uu = L D L' here (unit L): the factor sweep leaves Mbar = D^-1 L^-1 X, L^-1 and the four 1 / d_a; with m0 = L^-1 g the
    // corrector's costate is pi_k = Abar' pi - Mbar' m0 and its feed-forward term is m = D^-1 m0 (column 15 of Mbar).
    //
    // start point of the iteration (HPIPM-style cold start, oracle ocpqp_ipm): inputs pushed inside the box, lam = mu0 / slack
    auto init_point = [&](bool mine) {
        constexpr int CHI = 10;
        for (int k0 = 0; k0 < N; k0 += CHI) {
            T ulv[CHI];
            NMPC_UNROLL for (int i = 0; i < CHI; i++) ulv[i] = ulin((k0 + i < N) ? k0 + i : N - 1, ta);
            NMPC_UNROLL for (int i = 0; i < CHI; i++) {
                const int k = k0 + i;
                const T lo = lb_a - ulv[i], hi = ub_a - ulv[i];
                T thr = c.thr0;
                if (c.thr0_rel * (hi - lo) > thr) thr = c.thr0_rel * (hi - lo);
                if (hi - lo < T(2) * thr) thr = T(0.5) * (hi - lo);
                T v = 0;
                if (v - lo < thr) v = lo + thr;
                if (hi - v < thr) v = hi - thr;
                if (k < N && tc == 0 && mine) {
                    T *ivk = tIV_own + k * IV_ROWS;
                    ivk[ta] = v;
                    st2(ivk + IVP_T + 2 * ta, v - lo, hi - v);
                    st2(ivk + IVP_L + 2 * ta, c.mu0 / (v - lo), c.mu0 / (hi - v));
                }
            }
        }
    };
    T rmaxE = 0, dmaxE = 0;        // sweep E: largest inverse step length (floor tau), largest |d| / box width
    struct OpsD { T mn[4], y, ri, ll, lu, tl, tu, ua; };
    // ================= sweep D: backward homogeneous solve of the corrector: g = dr + B'pi, m0 = L^-1 g, pi_k = Abar'pi - Mbar'm0
    auto sweepD = [&](auto) {
        T Aq0[4], Aq1z[4], Bt[4];
        if (SHARED) {            // as the factor sweep's tiles, without b and the homogeneous 1
            NMPC_UNROLL for (int kt = 0; kt < 4; kt++) {
                const int l = natR[kt] >= 0 ? natR[kt] : 0;
                const bool real = natR[kt] >= 0;
                const T a0 = sAd[l * 8 + tc], a1 = sAd[l * 8 + 4 + (tc < 3 ? tc : 0)], bb = sB[l * 4 + tc];
                Aq0[kt] = real ? a0 : T(0);
                Aq1z[kt] = (real && tc < 3) ? a1 : T(0);
                Bt[kt] = real ? bb : T(0);
            }
        } else fetch_stage(N - 1, r);
        T pit[4];
        NMPC_UNROLL for (int t = 0; t < 4; t++) pit[t] = 0;
        const int kl = LDSC ? (lstg < N ? lstg : N) : 0;
        auto fetch_sc = [&](int kq, OpsD &o) {
            const int k = kq > 0 ? kq : 0;
            const T *ivn = tIV + k * IV_ROWS;
            { const D2 p_ = ld2(ivn + IVP_L + 2 * ta); o.ll = p_.x; o.lu = p_.y; } { const D2 p_ = ld2(ivn + IVP_T + 2 * ta); o.tl = p_.x; o.tu = p_.y; } o.ua = ivn[12 + ta];
        };
        auto fetch_d = [&](int kq, OpsD &o) {           // factors from the HBM scratch (clamped index, see sweep B)
            const int k = kq > 0 ? kq : 0;
            const T *lmn = tLM + k * TLM_ROWS;
            NMPC_UNROLL for (int jt = 0; jt < 4; jt++) o.mn[jt] = lmn[TLM_MT + jt * 16 + rT];     // Mbar[a][4jt+c]
            o.y = lmn[TLM_Z + rT];                                                             // (L^-1)[c][a]
            o.ri = lmn[TLM_RINV + ta];
            fetch_sc(kq, o);
        };
        auto fetch_d_lds = [&](int k, OpsD &o) {
            const T *lmn = sLM + k * LMR;
            NMPC_UNROLL for (int jt = 0; jt < 4; jt++) o.mn[jt] = lmn[jt * 16 + rT];
            o.y = lmn[64 + rT];
            o.ri = lmn[80 + ta];
        };
        auto stageD = [&](int k, OpsD &o, auto lds_tag) {
            constexpr bool LDSST = decltype(lds_tag)::value;
            if (!SHARED) {
                NMPC_UNROLL for (int kt = 0; kt < 4; kt++) { Aq0[kt] = pfs[kt * 3]; Aq1z[kt] = tc < 3 ? pfs[kt * 3 + 1] : T(0); Bt[kt] = pfs[kt * 3 + 2]; }
                if (k > 0) fetch_stage(k - 1, r);
            }
            T drt;
            {
                const Pair<T> pr(o.ll, o.lu, o.tl, o.tu);
                const T da = o.ua;                                           // the affine step
                const T dla = -o.ll - pr.kl * da, dua = -o.lu + pr.ku * da;
                const T cl = dla * da, cu = -dua * da;
                drt = tc == 0 ? -(sigmu - cl) * pr.itl + (sigmu - cu) * pr.itu : T(0);
            }
            const T g = mfma44(Bt[2], pit[2], mfma44(Bt[0], pit[0], drt)) + mfma44(Bt[3], pit[3], mfma44(Bt[1], pit[1], T(0)));
            const T m0 = mfma44(o.y, g, T(0));                        // lane (a,0): (L^-1 g)_a
            const T mst = o.ri * m0;
            if (LDSST) sLM[k * LMR + (tc == 0 ? 60 + ta : 84 + ta)] = mst;
            else tLM[k * TLM_ROWS + (tc == 0 ? TLM_MT + 60 + ta : TLM_RINV + 4 + ta)] = mst;
            if (k > 0) {
                T an[4];
                an[0] = pit[0];
                an[1] = dt_v * pit[0] + pit[1];
                an[2] = mfma44(Aq0[2], pit[2], mfma44(Aq0[0], pit[0], T(0))) + mfma44(Aq0[1], pit[1], T(0));       // (tile (3,0) of Abar is zero)
                an[3] = mfma44(Aq1z[2], pit[2], mfma44(Aq1z[0], pit[0], T(0))) + mfma44(Aq1z[3], pit[3], mfma44(Aq1z[1], pit[1], T(0)));
                NMPC_UNROLL for (int t = 0; t < 4; t++) {
                    const T v = an[t] - mfma44(o.mn[t], m0, T(0));
                    pit[t] = (tc == 0 && natR[t] >= 0) ? v : T(0);
                }
            }
        };
        using Fl = std::integral_constant<bool, false>;
        using Tr = std::integral_constant<bool, true>;
        constexpr int H = 2;
        OpsD oa[H], ob[H];
        NMPC_UNROLL for (int i = 0; i < H; i++) fetch_d(N - 1 - i, oa[i]);
        for (int k0 = N - 1; k0 >= kl; k0 -= 2 * H) {
            NMPC_UNROLL for (int i = 0; i < H; i++) fetch_d(k0 - H - i, ob[i]);
            NMPC_UNROLL for (int i = 0; i < H; i++) { if (k0 - i >= kl) stageD(k0 - i, oa[i], Fl{}); }
            NMPC_UNROLL for (int i = 0; i < H; i++) fetch_d(k0 - 2 * H - i, oa[i]);
            NMPC_UNROLL for (int i = 0; i < H; i++) { if (k0 - H - i >= kl) stageD(k0 - H - i, ob[i], Fl{}); }
        }
        if constexpr (LDSC) {
            if (kl > 0) {
                OpsD ol, on, on2;                          // scalars two stages ahead, as in sweep E
                fetch_sc(kl - 1, on);
                fetch_sc(kl - 2, on2);
                OpsD ol2;
                auto scalars = [&](OpsD &o, int k) {
                    o.ll = on.ll; o.lu = on.lu; o.tl = on.tl; o.tu = on.tu; o.ua = on.ua;
                    on.ll = on2.ll; on.lu = on2.lu; on.tl = on2.tl; on.tu = on2.tu; on.ua = on2.ua;
                    fetch_sc(k - 2, on2);
                };
                fetch_d_lds(kl - 1, ol);
                for (int k = kl - 1; k >= 0; k -= 2) {        // factors one stage ahead, two operand sets (as sweep B)
                    if (k >= 1) fetch_d_lds(k - 1, ol2);
                    scalars(ol, k);
                    stageD(k, ol, Tr{});
                    if (k >= 1) {
                        if (k >= 2) fetch_d_lds(k - 2, ol);
                        scalars(ol2, k - 1);
                        stageD(k - 1, ol2, Tr{});
                    }
                }
            }
        }
    };
    // ================= sweep E: forward homogeneous solve of the corrector, final direction, step length
    auto sweepE = [&](auto) {
        T AT2[4], AT3z[4], BT[4];
        if (SHARED) {
            NMPC_UNROLL for (int it = 0; it < 4; it++) {
                const int l = natC[it] >= 0 ? natC[it] : 0;
                const bool real = natC[it] >= 0;
                const T a2 = sAd[l * 8 + ta], a3 = sAd[l * 8 + 4 + (ta < 3 ? ta : 0)], bb = sB[l * 4 + ta];
                AT2[it] = real ? a2 : T(0);
                AT3z[it] = (real && ta < 3) ? a3 : T(0);
                BT[it] = real ? bb : T(0);
            }
        }
        T xt[4];
        NMPC_UNROLL for (int t = 0; t < 4; t++) xt[t] = 0;
        const T one15 = (ta == 3 && tc == 0) ? T(1) : T(0);        // homogeneous coordinate, for the m term only
        T rmx = c.tau, dmx = 0;
        struct OpsE { T mt[4], z, ll, lu, tl, tu, ua, ab[SHARED ? 1 : 12]; };
        const int kl = LDSC ? (lstg < N ? lstg : N) : 0;
        const int dslot = tc == 0 ? 16 + ta : 20 + ta;
        auto fetch_sc = [&](int k, OpsE &o) {
            const T *ivn = tIV + k * IV_ROWS;
            { const D2 p_ = ld2(ivn + IVP_L + 2 * ta); o.ll = p_.x; o.lu = p_.y; } { const D2 p_ = ld2(ivn + IVP_T + 2 * ta); o.tl = p_.x; o.tu = p_.y; } o.ua = ivn[12 + ta];
        };
        auto fetch_e = [&](int kq, OpsE &o) {
            const int k = kq < N ? kq : N - 1;
            const T *lmn = tLM + k * TLM_ROWS;
            if (SHARED || k >= kl) {
                NMPC_UNROLL for (int jt = 0; jt < 4; jt++) o.mt[jt] = lmn[TLM_MT + jt * 16 + r];
                o.z = lmn[TLM_Z + r];
            }
            fetch_sc(k, o);
            if (!SHARED) {
                const T *a = tAB + (size_t)k * TAB_ROWS + rT;
                NMPC_UNROLL for (int t = 0; t < 12; t++) o.ab[t] = a[t * 16];
            }
        };
        auto fetch_e_lds = [&](int k, OpsE &o) {
            const T *lmn = sLM + k * LMR;
            NMPC_UNROLL for (int jt = 0; jt < 4; jt++) o.mt[jt] = lmn[jt * 16 + r];
            o.z = lmn[64 + r];
        };
        auto stageE = [&](int k, OpsE &o) {
            if (!SHARED) {
                NMPC_UNROLL for (int it = 0; it < 4; it++) { AT2[it] = o.ab[it * 3]; AT3z[it] = ta < 3 ? o.ab[it * 3 + 1] : T(0); BT[it] = o.ab[it * 3 + 2]; }
                if (LDSC) { if (k < kl) fetch_e_lds(k, o); }
            }
            T *ivk = tIV + k * IV_ROWS;
            T xn[4];
            xn[0] = xt[0] + dt_v * xt[1]; xn[1] = xt[1]; xn[2] = 0; xn[3] = 0;
            NMPC_UNROLL for (int it = 0; it < 4; it++) xn[it] = mfma44(AT3z[it], xt[3], it < 3 ? mfma44(AT2[it], xt[2], xn[it]) : xn[it]);
            const T v = mfma44(o.mt[2], xt[2], mfma44(o.mt[0], xt[0], T(0)))
                      + mfma44(o.mt[3], xt[3] + one15, mfma44(o.mt[1], xt[1], T(0)));
            const T ut = mfma44_na(o.z, v, T(0));
            NMPC_UNROLL for (int it = 0; it < 4; it++) xn[it] = mfma44(BT[it], ut, xn[it]);
            {
                const Pair<T> pr(o.ll, o.lu, o.tl, o.tu);
                const T da = o.ua;
                const T dla = -o.ll - pr.kl * da, dua = -o.lu + pr.ku * da;
                const T cl = dla * da, cu = -dua * da;
                const T d = da + ut;
                ivk[dslot] = d;
                const T dl = -o.ll - (cl - sigmu) * pr.itl - pr.kl * d;
                const T du = -o.lu - (cu - sigmu) * pr.itu + pr.ku * d;
                rmx = fmax(rmx, fmax(-d * pr.itl, d * pr.itu));
                rmx = fmax(rmx, fmax(-dl * fast_rcp(o.ll), -du * fast_rcp(o.lu)));
                dmx = fmax(dmx, fabs(d) * iw_a);
            }
            NMPC_UNROLL for (int t = 0; t < 4; t++) xt[t] = xn[t];
        };
        constexpr int H = 2;
        OpsE oa[H], ob[H];
        const int kb = SHARED ? kl : 0;
        NMPC_UNROLL for (int i = 0; i < H; i++) fetch_e(kb + i, oa[i]);
        if constexpr (LDSC && SHARED) {
            if (kl > 0) {
                // the iterate's scalars come from global memory two stages ahead (a stage is shorter than an L2 round trip)
                OpsE ol, on, on2;
                fetch_sc(0, on);
                fetch_sc(1 < N ? 1 : 0, on2);
                OpsE ol2;
                auto scalars = [&](OpsE &o, int k) {
                    o.ll = on.ll; o.lu = on.lu; o.tl = on.tl; o.tu = on.tu; o.ua = on.ua;
                    on.ll = on2.ll; on.lu = on2.lu; on.tl = on2.tl; on.tu = on2.tu; on.ua = on2.ua;
                    fetch_sc(k + 2 < N ? k + 2 : N - 1, on2);
                };
                fetch_e_lds(0, ol);
                for (int k = 0; k < kl; k += 2) {            // factors of stage k + 1 out of LDS before stage k computes (as sweep B)
                    if (k + 1 < kl) fetch_e_lds(k + 1, ol2);
                    scalars(ol, k);
                    stageE(k, ol);
                    if (k + 1 < kl) {
                        if (k + 2 < kl) fetch_e_lds(k + 2, ol);
                        scalars(ol2, k + 1);
                        stageE(k + 1, ol2);
                    }
                }
            }
        }
        for (int k0 = kb; k0 < N; k0 += 2 * H) {
            NMPC_UNROLL for (int i = 0; i < H; i++) fetch_e(k0 + H + i, ob[i]);
            NMPC_UNROLL for (int i = 0; i < H; i++) { if (k0 + i < N) stageE(k0 + i, oa[i]); }
            NMPC_UNROLL for (int i = 0; i < H; i++) fetch_e(k0 + 2 * H + i, oa[i]);
            NMPC_UNROLL for (int i = 0; i < H; i++) { if (k0 + H + i < N) stageE(k0 + H + i, ob[i]); }
        }
        rmaxE = rmx; dmaxE = dmx;
    };
    // ================= sweep F: primal-dual update, duality measure of the new iterate, active-set guess for a later attempt
    T msF = 0;
    auto sweepF = [&](T alpha) {
        // element-wise: lane (a,c) takes input a of the stages k = c, c + 4, c + 8, ... (every lane works: four stages per
        // instruction instead of one replicated four times - the sweep is all FP64 vector arithmetic)
        T ms = 0;
        constexpr int CHF = 5;
        for (int k0 = 0; k0 < N; k0 += 4 * CHF) {
            T f_u[CHF], f_ll[CHF], f_lu[CHF], f_tl[CHF], f_tu[CHF], f_ua[CHF], f_d[CHF];
            NMPC_UNROLL for (int i = 0; i < CHF; i++) {
                const int kq = k0 + 4 * i + tc;
                const int k = kq < N ? kq : N - 1;
                const T *ivn = tIV + k * IV_ROWS;
                f_u[i] = ivn[ta]; { const D2 p_ = ld2(ivn + IVP_L + 2 * ta); f_ll[i] = p_.x; f_lu[i] = p_.y; } { const D2 p_ = ld2(ivn + IVP_T + 2 * ta); f_tl[i] = p_.x; f_tu[i] = p_.y; }
                f_ua[i] = ivn[12 + ta]; f_d[i] = ivn[16 + ta];
            }
            NMPC_UNROLL for (int i = 0; i < CHF; i++) {
                const int kq = k0 + 4 * i + tc;
                const bool live = kq < N;
                // a lane past the horizon repeats stage N - 1 into the spare slots of that stage
                T *ivk = tIV + (live ? kq : N - 1) * IV_ROWS;
                T u = f_u[i], ll = f_ll[i], lu = f_lu[i], tl = f_tl[i], tu = f_tu[i];
                const Pair<T> pr(ll, lu, tl, tu);
                const T da = f_ua[i], d = f_d[i];
                const T dla = -ll - pr.kl * da, dua = -lu + pr.ku * da;
                const T cl = dla * da, cu = -dua * da;
                const T dl = -ll - (cl - sigmu) * pr.itl - pr.kl * d;
                const T du = -lu - (cu - sigmu) * pr.itu + pr.ku * d;
                u += alpha * d; tl += alpha * d; tu -= alpha * d; ll += alpha * dl; lu += alpha * du;
                // (a lane past the horizon stores into the spare slots 20..23 of stage N - 1)
                ivk[live ? ta : 20 + ta] = u;
                st2(ivk + (live ? IVP_L + 2 * ta : 20 + 2 * (ta & 1)), ll, lu);
                st2(ivk + (live ? IVP_T + 2 * ta : 20 + 2 * (ta & 1)), tl, tu);
                // active-set guess for a later attempt: a bound whose multiplier exceeds its slack
                ivk[live ? 16 + ta : 20 + ta] = ll > tl ? T(-1) : (lu > tu ? T(1) : T(0));
                ms += live ? ll * tl + lu * tu : T(0);
            }
        }
        msF = ms;
    };
    // ================= warm start of the interior point: the last pass of an exhausted attempt (slots 24..35) becomes the iterate
    // (slots 0..11); duality measure of that point.  Element-wise, stages spread over the lanes as in sweep F.
    auto sweepW = [&](auto) {
        T ms = 0;
        constexpr int CHW = 5;
        for (int k0 = 0; k0 < N; k0 += 4 * CHW) {
            T w_u[CHW], w_l[CHW], w_h[CHW], w_ul[CHW];
            NMPC_UNROLL for (int i = 0; i < CHW; i++) {
                const int kq = k0 + 4 * i + tc;
                const int k = kq < N ? kq : N - 1;
                const T *ivn = tIV + k * IV_ROWS;
                w_u[i] = ivn[24 + ta]; w_l[i] = ivn[28 + ta]; w_h[i] = ivn[32 + ta]; w_ul[i] = ulin(k, ta);
            }
            NMPC_UNROLL for (int i = 0; i < CHW; i++) {
                const int kq = k0 + 4 * i + tc;
                const bool live = kq < N;
                T *ivk = tIV + (live ? kq : N - 1) * IV_ROWS;
                ivk[live ? ta : 20 + ta] = w_u[i];
                st2(ivk + (live ? IVP_L + 2 * ta : 20 + 2 * (ta & 1)), w_l[i], w_h[i]);
                const T lo = lb_a - w_ul[i], hi = ub_a - w_ul[i];
                const T tl = w_u[i] - lo, tu = hi - w_u[i];                  // the slacks of the new iterate start on its inputs
                st2(ivk + (live ? IVP_T + 2 * ta : 20 + 2 * (ta & 1)), tl, tu);
                ms += live ? w_l[i] * tl + w_h[i] * tu : T(0);
            }
        }
        msF = ms;
    };
    // ================= state rollout from the inputs of the iterate (an instance that ends on the interior-point iterate): xhat_k
    // for the output sweep, NaN check of the step
    bool roll_bad = false;
    auto rollout = [&](auto) {
        T AT2[4], AT3[4], BT[4];
        auto tiles_T = [&](int k) {
            if (SHARED) {
                NMPC_UNROLL for (int it = 0; it < 4; it++) {
                    const int l = natC[it] >= 0 ? natC[it] : 0;
                    const bool real = natC[it] >= 0;
                    const T a2 = sAd[l * 8 + ta], a3 = sAd[l * 8 + 4 + (ta < 3 ? ta : 0)], bb = sB[l * 4 + ta], bv_ = sbv[l];
                    AT2[it] = real ? a2 : T(0);
                    AT3[it] = real ? (ta < 3 ? a3 : bv_) : ((it == 3 && tc == 3 && ta == 3) ? T(1) : T(0));
                    BT[it] = real ? bb : T(0);
                }
            } else {
                const T *a = tAB + (size_t)k * TAB_ROWS + rT;
                NMPC_UNROLL for (int it = 0; it < 4; it++) { AT2[it] = a[(it * 3) * 16]; AT3[it] = a[(it * 3 + 1) * 16]; BT[it] = a[(it * 3 + 2) * 16]; }
            }
        };
        if (SHARED) tiles_T(0);
        T xt[4];
        NMPC_UNROLL for (int t = 0; t < 4; t++) xt[t] = (t == 3 && ta == 3 && tc == 0) ? T(1) : T(0);
        int xslot[4];
        NMPC_UNROLL for (int t = 0; t < 4; t++) xslot[t] = (tc == 0 && natR[t] >= 0) ? natR[t] : 13;
        bool bad = false;
        for (int k = 0; k < N; k++) {
            if (!SHARED) tiles_T(k);
            const T uk = tIV[k * IV_ROWS + ta];
            const T ut = tc == 0 ? uk : T(0);
            bad |= !(uk == uk) || fabs(uk) > T(1e300);
            if (TRAJ) {
                T *xs = tLM + k * TLM_ROWS + 66;
                NMPC_UNROLL for (int t = 0; t < 4; t++) xs[xslot[t]] = xt[t];
            }
            T xn[4];
            xn[0] = xt[0] + dt_v * xt[1]; xn[1] = xt[1]; xn[2] = 0; xn[3] = 0;
            NMPC_UNROLL for (int it = 0; it < 4; it++) xn[it] = mfma44(AT3[it], xt[3], it < 3 ? mfma44(AT2[it], xt[2], xn[it]) : xn[it]);
            NMPC_UNROLL for (int it = 0; it < 4; it++) xn[it] = mfma44(BT[it], ut, xn[it]);
            NMPC_UNROLL for (int t = 0; t < 4; t++) { xt[t] = xn[t]; bad |= (tc == 0 && natR[t] >= 0) && (!(xn[t] == xn[t]) || fabs(xn[t]) > T(1e300)); }
        }
        if (tc == 0) {
            NMPC_UNROLL for (int t = 0; t < 4; t++)
                if (natR[t] >= 0) sXh[natR[t]] = xt[t];
        }
        NMPC_WSYNC();
        const T xN = sXh[rr];
        if (TRAJ) { if (rowl) tLM[66 + rr] = xN; }
        roll_bad = bad;
    };

    // verdict of an external factorisation (MODE 3): usable | NaN pivot | team-wide max |B'PB|
    bool ext_ok = true, ext_nan = false;
    T ext_gm = 0;
    // ================= one active-set pass of every team of the wave that is in an attempt (mode M_POL)
    auto as_pass = [&]() {
        pol = mode == M_POL;
        tLM = pol ? tLM_own : tLM_spare; tIV = pol ? tIV_own : tIV_spare; tP = pol ? tP_own : tP_spare;
        // the wave sweeps from the highest stage any of its live teams needs (wave-uniform trip count)
        ks = N - 1;
        if (have_tP && pass > 0) {
            if (r == 0) { sRed[28] = (T)(pol ? k_top : -1); sRed[29] = (T)(pol ? ck_valid : N); }
            __syncthreads();
            ks = 0;
            int vmin = N;            // every live team must own the checkpoint the wave resumes from
            for (int t = 0; t < 4; t++) {
                const int kt = (int)smem[t * LDS_T + A_RED + 28], vt = (int)smem[t * LDS_T + A_RED + 29];
                ks = kt > ks ? kt : ks;
                vmin = vt < vmin ? vt : vmin;
            }
            if (ks >= vmin) ks = N - 1;
            ks = __builtin_amdgcn_readfirstlane(ks);
        }
        // checkpoint window of this pass: two stages in the first pass, the configured window after
        wnd = pass_in_attempt == 0 ? (ckpt < 2 ? ckpt : 2) : ckpt;
        if (pol) ck_valid = (wnd < ks) ? wnd : ck_valid;

        ok = true; nanp = false; gm = 0;
        T gt;
        if constexpr (MODE == 3) {
            ok = ext_ok; nanp = ext_nan; gt = ext_gm;    // the factorisation came from the block-parallel launches
        } else {
            if (nopins_pass) sweepA(NoPins{}, NoIpm{}); else sweepA(WithPins{}, NoIpm{});
            NMPC_STAMP(0)
            __syncthreads();
            // growth certificate: the team-wide max |B'PB| of this sweep against that of the first factorisation of the solve
            sh[r] = gm;
            NMPC_WSYNC();
            gt = sh[0];
            NMPC_UNROLL for (int i = 1; i < 16; i++) gt = fmax(gt, sh[i]);
        }
        if (pol && gbase == T(0)) gbase = gt;
        const bool trip = pol && c.growth_max > T(0) && gt > c.growth_max * gbase;
        bool pol_fail = false;
        if (pol && !ok) {
            // An invalid pivot (NaN, out of range, not positive) is one more way for the attempt to fail: the interior-point iteration takes
            // over, as in the oracle (ocpqp_polish breaks out of such a pass) - with NaN data it fails at its first factorisation, and the
            // end of the kernel classes the failure by the inputs.  (Until late round 5 a NaN pivot of the FIRST pass ended the solve here
            // with status 1 and no iteration, where the oracle counts the one the interior point attempts: draw 431.)  In a later pass it can
            // be the pinned recursion itself - a long saturated stretch of an unstable plant is an open loop, P grows by rho(A)^2 per stage
            pol_fail = true;                            // give up this attempt
        }
        if (trip) { pol_fail = true; tripped = true; }  // not accurate enough to be accepted (see qp_growth_max): the attempt fails
        pol2 = mode == M_POL;

        viol = false; heavy = false; kchgB = -1; xh = 0; anyp = false;
        // the last pass of the attempt (wave-uniform: the live teams of a wave entered the attempt together)
        // ... and a later attempt is still allowed: the warm start is for the interior-point iteration BETWEEN two attempts
        const bool warm_ok = pass_in_attempt == polish_passes - 1 && npol - pass_in_attempt + polish_passes < c.polish_budget;
        const bool lastp = c.warm_start != 0 && __ballot(pol2 && warm_ok) != 0;
        if (MODE == 3 && tcx.frec) {
            // the blocks of the horizon swept forward at the same time (phase 3): their records stand in for the sweep
            const T *fr = tcx.frec + (size_t)(pol2 ? (size_t)inst : (size_t)w.Bp) * tcx.J * FR_ROWS;
            int kcm = -1;
            for (int b = 0; b < tcx.J; b++) {
                viol |= fr[b * FR_ROWS] != T(0); heavy |= fr[b * FR_ROWS + 1] != T(0); anyp |= fr[b * FR_ROWS + 3] != T(0);
                const int kb_ = (int)fr[b * FR_ROWS + 2];
                kcm = kb_ > kcm ? kb_ : kcm;
            }
            u0_cand = pol2 ? fr[4 + ta] : u0_cand;
            xh = fr[(tcx.J - 1) * FR_ROWS + 8 + rr];
            if (tc == 0) sRed[28 + ta] = (T)kcm;
        }
        else if (nopins_pass) sweepB(NoPins{}, NoIpm{}, NoWarm{});
        else if (lastp) sweepB(WithPins{}, NoIpm{}, Warm{});
        else sweepB(WithPins{}, NoIpm{}, NoWarm{});
        NMPC_STAMP(1)
        if (TRAJ) { if (rowl) tLM[66 + rr] = xh; }      // xhat_N parks in the unused xhat_0 slot (output sweep)
        __syncthreads();
        // team-wide flags from wave-wide ballots (only lanes (a,0) carry inputs; every row of xhat_N is checked)
        const bool t_viol = (__ballot(viol && tc == 0) & team_mask) != 0;
        const bool t_heavy = (__ballot((heavy && tc == 0) || !(xh == xh)) & team_mask) != 0;
        const int kc = (int)fmax(fmax(sRed[28], sRed[29]), fmax(sRed[30], sRed[31]));
        const bool t_anyp = (__ballot(anyp && tc == 0) & team_mask) != 0;
        const bool warm_ok_prev = warm_ok;
        if (pol2) {
            npol++;
            pass_in_attempt++;
            const bool unclean = pol_fail || t_viol || t_heavy;
            if (!unclean) { mode = M_DONE; from_ua = true; }   // the pass satisfies the KKT conditions of the QP: accepted
            else if (pol_fail || t_heavy || pass_in_attempt >= polish_passes) {
                mode = M_GIVEUP;
                // out of passes after a finished pass with pins: that pass seeds the interior-point iteration
                warm_avail = lastp && warm_ok_prev && !nopins_pass && !pol_fail && !t_heavy && t_anyp;
            }
            else k_top = kc < ck_valid ? kc : N - 1;
        }
        pass++;
        __syncthreads();   // sRed / sXh are reused by the next pass
    };

    int it = 0;              // interior-point iterations taken
    bool tail_fin = false;   // MODE 3: this step ended the instance on an accepted pass
    if constexpr (MODE == 0) {
        for (;;) {
            if (__ballot(mode == M_POL) == 0) break;
            if (tcx.cap > 0 && pass >= tcx.cap) break;       // long horizon: the block-parallel tail continues the attempt
            nopins_pass = pass == 0;
            as_pass();
        }
    } else if constexpr (MODE == 3) {
        // ---- one step of the block-parallel tail.  The factorisation of this step was done by the launches
        // of nmpc_block.hpp; what remains of an interior-point iteration (phase 1) or of an active-set pass (phase 2) runs here, and
        // the team's state travels in its tail-state row.  The tail covers the common continuation of a long-horizon instance - one
        // interior-point iteration from the warm start, then one more attempt - with the same sweeps as MODE 2; anything else (no warm
        // start, a factorisation that cannot be used, a second iteration, an attempt that fails) goes to the fallback list, which
        // k_team_qp_list solves from the active-set kernel's hand-over as if the tail had not run.
        const T nc = T(2 * NU) * T(N);
        T *tsr = tcx.ts + (size_t)winst * TS_ROWS;
        int tstate = valid ? (int)tsr[0] : (int)TS_NONE;
        T mu = tsr[4], rho = tsr[5], pol_mu = tsr[7], step_last = tsr[8];
        npol = (int)tsr[1]; pass_in_attempt = (int)tsr[2]; gbase = tsr[3]; it = (int)tsr[6];
        if (tcx.phase != 0) {
            const T *bi = tcx.binfo + (size_t)winst * tcx.J * 2;
            for (int b = 0; b < tcx.J; b++) {
                const T fl = bi[2 * b + 1];
                ext_gm = fmax(ext_gm, bi[2 * b]); ext_ok &= fl == T(0); ext_nan |= fl == T(2);
            }
        }
        mode = M_DONE;
        if (tcx.phase == 0) {
            const bool mid = tstate == TS_AS;            // handed over in the middle of its first attempt: the row is complete
            const int np0 = w.npol[inst];
            const T gb0 = w.gbase[inst];
            if (!mid) { npol = np0 < 0 ? -np0 : np0; pass_in_attempt = 0; gbase = fabs(gb0); }
            pol_mu = c.polish_mu * (mid ? T(1) : T(1e-2));
            warm_avail = valid && !mid && gb0 < T(0);
            mu = c.mu0; rho = T(1); step_last = 0;
            if (__ballot(warm_avail) != 0) {
                tIV = warm_avail ? tIV_own : tIV_spare;
                sweepW(Ipm{});
                sh[r] = msF;
                __syncthreads();
                T ms = 0;
                NMPC_UNROLL for (int i = 0; i < 16; i++) ms += sh[i];
                mu = ms / nc;
                __syncthreads();
            }
            tstate = valid ? (mid ? (int)TS_AS : (warm_avail ? (int)TS_IPM : (int)TS_FALLBACK)) : (int)TS_NONE;
            warm_avail = false;
            it = mid ? 0 : 1;     // the iteration the next launches perform
        } else if (tcx.phase == 1) {
            const bool ipm = tstate == TS_IPM;
            if (__ballot(ipm) != 0) {
                const bool trip = c.growth_max > T(0) && !(ext_gm <= c.growth_max * gbase);
                const bool ipm2 = ipm && ext_ok && !trip;
                pol = false; pol2 = false;
                tLM = ipm2 ? tLM_own : tLM_spare; tIV = ipm2 ? tIV_own : tIV_spare; tP = tP_spare;
                viol = false; heavy = false; kchgB = -1; xh = 0;
                sweepB(NoPins{}, Ipm{}, NoWarm{});
                __syncthreads();
                {
                    const T rmax = fmax(fmax(sRed[4], sRed[5]), fmax(sRed[6], sRed[7]));
                    const T s2 = sRed[8] + sRed[9] + sRed[10] + sRed[11];
                    const T aaff = T(1) / rmax;
                    const T muaff = (T(1) - aaff) * mu + aaff * aaff * s2 / nc;
                    T sg3 = muaff / mu;
                    sg3 = sg3 * sg3 * sg3;
                    sigmu = sg3 * mu;
                }
                sweepD(Ipm{});
                __syncthreads();
                sweepE(Ipm{});
                if (tc == 0) { sRed[12 + ta] = rmaxE; sRed[20 + ta] = dmaxE; }
                __syncthreads();
                const T rmx = fmax(fmax(sRed[12], sRed[13]), fmax(sRed[14], sRed[15]));
                const T dmx = fmax(fmax(sRed[20], sRed[21]), fmax(sRed[22], sRed[23]));
                const T alpha = c.tau / rmx;
                sweepF(alpha);
                sh[r] = msF;
                __syncthreads();
                T ms = 0;
                NMPC_UNROLL for (int i = 0; i < 16; i++) ms += sh[i];
                const bool good = ipm2 && alpha == alpha && !(alpha < T(1e-12));
                if (good) { rho *= (T(1) - alpha); mu = ms / nc; step_last = alpha * dmx; }
                // what the interior point does next (top of its loop in the other modes): the tail follows it into an attempt only
                const bool conv = mu <= c.tol_comp && rho <= c.tol_stat && (!(c.tol_step > T(0)) || step_last <= c.tol_step);
                const bool attempt = c.polish && mu <= pol_mu && npol < c.polish_budget;
                if (ipm) tstate = (good && mu == mu && !conv && attempt) ? (int)TS_AS : (int)TS_FALLBACK;
                pass_in_attempt = 0;
                __syncthreads();
            }
        } else if (tcx.phase == 3) {
            // forward sweep of block tcx.blk of an active-set pass (the blocks of an instance at the same time, from the states the
            // boundary scan left): KKT check, pin codes, candidate inputs and - last pass of a first attempt - the warm start of
            // its stages; what the decision of the pass needs goes to the block's record
            mode = tstate == TS_AS ? M_POL : M_DONE;
            if (__ballot(mode == M_POL) != 0) {
                pol = mode == M_POL; pol2 = pol;
                tLM = pol ? tLM_own : tLM_spare; tIV = pol ? tIV_own : tIV_spare; tP = tP_spare;
                nopins_pass = false;
                fwd_s = tcx.blk * tcx.M; fwd_e = fwd_s + tcx.M < N ? fwd_s + tcx.M : N;
                fwd_x = tcx.xb + ((size_t)winst * tcx.J + tcx.blk) * 16;
                viol = false; heavy = false; kchgB = -1; xh = 0; anyp = false;
                const bool warm_ok = pass_in_attempt == polish_passes - 1 && npol - pass_in_attempt + polish_passes < c.polish_budget;
                const bool lastp = c.warm_start != 0 && __ballot(pol2 && warm_ok) != 0;
                if (lastp) sweepB(WithPins{}, NoIpm{}, Warm{});
                else sweepB(WithPins{}, NoIpm{}, NoWarm{});
                __syncthreads();
                const bool t_viol = (__ballot(viol && tc == 0) & team_mask) != 0;
                const bool t_heavy = (__ballot(heavy && tc == 0) & team_mask) != 0;
                const bool t_anyp = (__ballot(anyp && tc == 0) & team_mask) != 0;
                const int kc = (int)fmax(fmax(sRed[28], sRed[29]), fmax(sRed[30], sRed[31]));
                T *fr = tcx.frec + ((size_t)winst * tcx.J + tcx.blk) * FR_ROWS;
                if (r == 0) { fr[0] = t_viol ? T(1) : T(0); fr[1] = t_heavy ? T(1) : T(0); fr[2] = (T)kc; fr[3] = t_anyp ? T(1) : T(0); }
                if (tc == 0) fr[4 + ta] = u0_cand;
                if (rowl) fr[8 + rr] = xh;
                __syncthreads();
            }
            mode = M_DONE;
        } else {
            mode = tstate == TS_AS ? M_POL : M_DONE;
            if (__ballot(mode == M_POL) != 0) {
                nopins_pass = false;
                k_top = N - 1; ck_valid = 0;
                const bool first_attempt = npol < polish_passes;      // (a second attempt starts with all passes of the first one spent)
                as_pass();
                // the first attempt ends in here when the first launch handed it over running: the hand-over the active-set kernel
                // would have written (the fallback list relies on it), and the warm start becomes the interior point's iterate
                const bool ended = tstate == TS_AS && mode == M_GIVEUP && first_attempt;
                if (ended && r == 0) {
                    w.npol[inst] = -(tripped ? c.polish_budget : npol);
                    w.gbase[inst] = warm_avail ? -gbase : gbase;
                }
                const bool install = ended && warm_avail;
                if (__ballot(install) != 0) {
                    __syncthreads();
                    tIV = install ? tIV_own : tIV_spare;
                    sweepW(Ipm{});
                    sh[r] = msF;
                    __syncthreads();
                    T ms = 0;
                    NMPC_UNROLL for (int i = 0; i < 16; i++) ms += sh[i];
                    if (install) { mu = ms / nc; rho = T(1); step_last = 0; it = 1; pol_mu = c.polish_mu * T(1e-2); pass_in_attempt = 0; }
                    __syncthreads();
                }
                if (tstate == TS_AS) tstate = mode == M_DONE ? (int)TS_DONE : (mode == M_POL ? (int)TS_AS : (install ? (int)TS_IPM : (int)TS_FALLBACK));
                tail_fin = tstate == TS_DONE && status == 0;
                if (tstate == TS_DONE && status != 0) tstate = TS_FALLBACK;
                warm_avail = false;
            }
            mode = M_DONE;
        }
        if (valid && r == 0 && tcx.phase != 3) {
            if (tstate == TS_FALLBACK) {
                const int slot = atomicAdd(tcx.fb_count, 1);
                tcx.fb_list[slot] = inst;
                tstate = TS_LISTED;
            }
            if (tstate == TS_AS || tstate == TS_IPM) {
                const int slot = atomicAdd(tcx.nx_count, 1);
                tcx.nx_list[slot] = inst;
            }
            tsr[0] = (T)tstate; tsr[1] = (T)npol; tsr[2] = (T)pass_in_attempt; tsr[3] = gbase; tsr[4] = mu; tsr[5] = rho;
            tsr[6] = (T)it; tsr[7] = pol_mu; tsr[8] = step_last;
            // the next pass of the same attempt re-aggregates only the blocks in which a pin code changed (nmpc_block.hip)
            tsr[10] = (tcx.phase == 2 && tcx.frec && tstate == TS_AS) ? T(1) : T(0);
        }
    } else {
        // ---- phases of a wave: interior-point iterations for the teams in that mode until each has converged, failed or
        // reached the threshold of its next attempt; then active-set passes for the teams that wait for an attempt (the
        // others idle in the spare row either way).  A team's own sequence of sweeps never depends on its wave-mates.
        const T nc = T(2 * NU) * T(N);
        T mu = c.mu0, rho = T(1), pol_mu = c.polish_mu, step_last = 0;
        bool have_point = false;
        bool just_attempted = MODE == 2;   // an attempt has just failed: one interior-point iteration before the next one (oracle ocpqp_ipm)
        const bool attempt_first = MODE == 1 && c.polish && c.polish_budget > 0 && c.polish_passes > 0 && c.polish_mu >= c.mu0;
        bool first_free = attempt_first; // wave-uniform: the first active-set phase starts from "all inputs free"
        if (MODE == 2) {                 // the first attempt was made - and given up - by the active-set kernel
            const int np0 = w.npol[inst];
            npol = np0 < 0 ? -np0 : np0;
            pol_mu *= T(1e-2);
            const T gb0 = w.gbase[inst];
            gbase = fabs(gb0);
            warm_avail = valid && gb0 < T(0);
            mode = valid ? M_IPM : M_DONE;
        } else {
            mode = valid ? (attempt_first ? M_WAIT : M_IPM) : M_DONE;
        }
        const int polish_budget = c.polish_budget, iter_max = c.iter_max > 0 ? c.iter_max : 1;
        // the interior point takes over from an attempt that ran out of passes: that attempt's last pass becomes its iterate
        auto install_warm = [&]() {
            if (__ballot(warm_avail) != 0) {
                tIV = warm_avail ? tIV_own : tIV_spare;
                sweepW(Ipm{});
                sh[r] = msF;
                __syncthreads();
                T ms = 0;
                NMPC_UNROLL for (int i = 0; i < 16; i++) ms += sh[i];
                if (warm_avail) { mu = ms / nc; rho = T(1); have_point = true; step_last = 0; warm_derived = true; }
                warm_avail = false;
                __syncthreads();
            }
        };
        install_warm();
        for (;;) {
            // ---------------- interior-point phase
            for (;;) {
                if (mode == M_IPM) {
                    if (!(mu == mu)) { status = 1; mode = M_DONE; }
                    else if (mu <= c.tol_comp && rho <= c.tol_stat && (it == 0 || !(c.tol_step > T(0)) || step_last <= c.tol_step)) mode = M_DONE;
                    else if (c.polish && mu <= pol_mu && npol < polish_budget && !just_attempted) mode = M_WAIT;
                    else if (it >= iter_max) { status = 2; mode = M_DONE; }
                }
                if (__ballot(mode == M_IPM) == 0) break;
                const bool ipm = mode == M_IPM;
                if (ipm) it++;
                if (__ballot(ipm && !have_point) != 0) {      // first interior-point iteration of some team
                    init_point(valid && ipm && !have_point);
                    __syncthreads();
                }
                have_point |= ipm;
                pol = false; pol2 = false;
                tLM = ipm ? tLM_own : tLM_spare; tIV = ipm ? tIV_own : tIV_spare; tP = tP_spare;
                ks = N - 1; wnd = -1;
                ok = true; nanp = false; gm = 0;
                sweepA(NoPins{}, Ipm{});
                NMPC_STAMP(0)
                __syncthreads();
                sh[r] = gm;
                NMPC_WSYNC();
                T gt = sh[0];
                NMPC_UNROLL for (int i = 1; i < 16; i++) gt = fmax(gt, sh[i]);
                if (ipm && ok && gbase == T(0)) gbase = gt;
                const bool trip = ipm && ok && c.growth_max > T(0) && !(gt <= c.growth_max * gbase);
                if (ipm && (!ok || trip)) {
                    // this factorisation cannot be used: the QP ends at the current iterate - solved if that is within the
                    // acceptable tolerances, a failure otherwise (oracle ocpqp_ipm)
                    status = (mu <= c.acc_comp && rho <= c.acc_stat) ? 0 : ((!ok && nanp) ? 1 : 4);
                    tripped |= trip;
                    mode = M_DONE;
                }
                const bool ipm2 = mode == M_IPM;
                tLM = ipm2 ? tLM_own : tLM_spare; tIV = ipm2 ? tIV_own : tIV_spare;
                viol = false; heavy = false; kchgB = -1; xh = 0;
                sweepB(NoPins{}, Ipm{}, NoWarm{});
                NMPC_STAMP(1)
                __syncthreads();
                {
                    const T rmax = fmax(fmax(sRed[4], sRed[5]), fmax(sRed[6], sRed[7]));
                    const T s2 = sRed[8] + sRed[9] + sRed[10] + sRed[11];
                    const T aaff = T(1) / rmax;
                    const T muaff = (T(1) - aaff) * mu + aaff * aaff * s2 / nc;
                    T sg3 = muaff / mu;
                    sg3 = sg3 * sg3 * sg3;
                    sigmu = sg3 * mu;
                }
                sweepD(Ipm{});
                NMPC_STAMP(2)
                __syncthreads();
                sweepE(Ipm{});
                NMPC_STAMP(3)
                if (tc == 0) { sRed[12 + ta] = rmaxE; sRed[20 + ta] = dmaxE; }
                __syncthreads();
                const T rmx = fmax(fmax(sRed[12], sRed[13]), fmax(sRed[14], sRed[15]));
                const T dmx = fmax(fmax(sRed[20], sRed[21]), fmax(sRed[22], sRed[23]));
                const T alpha = c.tau / rmx;
                sweepF(alpha);
                NMPC_STAMP(4)
                sh[r] = msF;
                __syncthreads();
                T ms = 0;
                NMPC_UNROLL for (int i = 0; i < 16; i++) ms += sh[i];
                if (ipm2) {
                    if (!(alpha == alpha)) { status = 1; mode = M_DONE; }
                    else if (alpha < T(1e-12)) { status = 3; mode = M_DONE; }
                    else {
                        rho *= (T(1) - alpha);
                        mu = ms / nc;
                        step_last = alpha * dmx;
                    }
                    just_attempted = false;
                }
                __syncthreads();   // sRed is reused by the next iteration
            }
            // ---------------- active-set phase: the teams that wait for an attempt
            if (__ballot(mode == M_WAIT) != 0) {
                if (mode == M_WAIT) { mode = M_POL; pass_in_attempt = 0; k_top = N - 1; ck_valid = 0; }
                pass = 0;
                for (;;) {
                    if (__ballot(mode == M_POL) == 0) break;
                    nopins_pass = first_free && pass == 0;
                    as_pass();
                }
                first_free = false;
                if (mode == M_GIVEUP) {       // the interior point iteration takes over (or resumes); next attempt 100x further down
                    mode = M_IPM;
                    pol_mu *= T(1e-2);
                    if (tripped) npol = polish_budget;
                    just_attempted = true;
                    if (!warm_avail && warm_derived && npol >= polish_budget) {
                        // no attempt is left and the iterate in hand descends from a warm start - off the central path, a poor
                        // place to converge from: the interior point finishes the QP from its standard cold point (oracle ocpqp_ipm)
                        have_point = false; mu = c.mu0; rho = T(1); warm_derived = false;
                    }
                }
                install_warm();
            }
            if (__ballot(mode != M_DONE) == 0) break;
        }
    }

    // ---- outputs.  Accepted active-set solution: the forward sweep left the candidate inputs (and xhat_k); interior-point
    // iterate: the inputs of the iterate, states by the rollout above.  The full step (U1) is applied stage-parallel into the
    // caller's arrays.  Failure (NaN: status 1, QP failure: 4, reported iteration cap: 2): zeros and the cold-start point.
    // Given up (MODE 0): the work-list launch continues the instance and writes everything.
    NMPC_STAMP(6)
    int nlp_status = status;
    if constexpr (MODE != 0) {
        // QP status -> acados numbering (oracle orc_sqp_rti): iteration cap tolerated or reported (U10 switch), min step -> QP failure
        nlp_status = status == 2 ? (c.maxiter_status ? 2 : 0) : (status == 3 ? 4 : status);
        const bool need_roll = MODE != 3 && valid && !from_ua && nlp_status == 0;
        if (__ballot(need_roll) != 0) {
            tLM = need_roll ? tLM_own : tLM_spare; tIV = need_roll ? tIV_own : tIV_spare;
            rollout(Ipm{});
            __syncthreads();
            const bool t_bad = (__ballot(roll_bad) & team_mask) != 0;
            if (need_roll && t_bad) nlp_status = 1;          // NaN anywhere in the step poisons the instance
        }
    }
    // The class of a failure (nmpc_ipm.hpp inputs_not_finite; oracle orc_sqp_rti): not-a-number data (1) exactly when one of the instance's own
    // inputs is not finite, a QP failure (4) otherwise - whatever arithmetic event ended the solve.  A rare path: the team scans its inputs again.
    if (__ballot(valid && (nlp_status == 1 || nlp_status == 4)) != 0) {
        bool nf = false;
        auto scan = [&](const TI *p, int n) { for (int i = r; i < n; i += 16) { const T v = (T)p[i]; nf |= !(fabs(v) <= T(1.7976931348623157e308)); } };
        scan(x0p, NX); scan(yr, N * NY); scan(ye, NX);
        if (in.x_init != nullptr && in.u_init != nullptr) {
            scan(in.x_init + (size_t)inst * (N + 1) * NX, (N + 1) * NX);
            scan(in.u_init + (size_t)inst * N * NU, N * NU);
        }
        const bool t_nf = (__ballot(nf) & team_mask) != 0;
        if (nlp_status == 1 || nlp_status == 4) nlp_status = t_nf ? 1 : 4;
    }
    NMPC_PROF_END(w)
    tLM = tLM_own; tIV = tIV_own;
    if (!valid) return false;
    if (MODE == 3 && !tail_fin) return false;    // the instance is still in the tail, or went to the fallback list
    if (MODE == 0 && mode == M_POL) {
        // the attempt is still running (pass cap of a long-horizon solve): its state goes to the tail's row - the pins of the next
        // pass are in the workspace already
        if (r == 0) {
            T *tsr = tcx.ts + (size_t)inst * TS_ROWS;
            tsr[0] = (T)TS_AS; tsr[1] = (T)npol; tsr[2] = (T)pass_in_attempt; tsr[3] = gbase;
            const int slot = atomicAdd(wl.count, 1);
            wl.list[slot] = inst;
        }
        return false;
    }
    if (MODE == 0 && mode == M_GIVEUP) {
        if (r == 0) {
            // the continuation (work-list launch, or MODE 2 on this wave) resumes the pass budget from here (all of it spent when the
            // growth certificate ended the attempt: the same pins would fail the same way) and compares against the same first factorisation
            w.npol[inst] = -(tripped ? c.polish_budget : npol);
            w.gbase[inst] = warm_avail ? -gbase : gbase;     // < 0: the attempt ran out of passes, its last pass seeds the interior point
            if (!tcx.inplace) {
                const int slot = atomicAdd(wl.count, 1);
                wl.list[slot] = inst;
            }
        }
        return tcx.inplace != 0;
    }
    const bool accepted = nlp_status == 0;
    const int uoff = from_ua ? 12 : 0;                 // candidate inputs of the accepted pass | inputs of the iterate
    if (r == 0) {
        if (out.status) out.status[inst] = nlp_status;
        w.iters[inst] = it; w.status[inst] = nlp_status; w.npol[inst] = (accepted && from_ua) ? npol : -npol;
    }
    if (tc == 0) {                                     // controller.py:448-452
        const T du0 = from_ua ? u0_cand : tIV[ta];
        out.u0[(size_t)inst * NU + ta] = (TI)(accepted ? ulin(0, ta) + du0 : T(0));
    }
    if (TRAJ) {
        constexpr int CH = 8;
        for (int k0 = 0; k0 <= N; k0 += CH) {
            T uv[CH], ulv[CH], xlv[CH], xhv[CH];
            NMPC_UNROLL for (int i = 0; i < CH; i++) {
                const int k = (k0 + i <= N) ? k0 + i : N, ku = k < N ? k : N - 1;
                uv[i] = tIV[ku * IV_ROWS + uoff + j];
                ulv[i] = ulin(ku, j);
                xlv[i] = xlin(k);
                xhv[i] = tLM[(k < N ? k * TLM_ROWS : 0) + 66 + rr];      // xhat_N sits in the stage-0 slot
            }
            NMPC_UNROLL for (int i = 0; i < CH; i++) {
                const int k = k0 + i;
                if (k <= N) {
                    if (out.x_out && rowl)
                        out.x_out[((size_t)inst * (N + 1) + k) * NX + rr] = (TI)(accepted ? xlv[i] + (k > 0 ? xhv[i] : T(0)) : x0r);
                    if (out.u_out && cmpl && k < N) out.u_out[((size_t)inst * N + k) * NU + j] = (TI)(accepted ? ulv[i] + uv[i] : T(0));
                }
            }
        }
    }
    return false;
}

// Body of k_team_as (built by two translation units: nmpc_as.hip and its default-codegen twin in nmpc_qp.hip).
// CONT builds (one wave per SIMD) continue a team whose first attempt failed AT ONCE, on the same wave: team_as MODE 2 - what
// k_team_qp_list runs for a work-list entry - from the hand-over record the attempt has just written.  No second launch then: on the
// headline workload that launch found an empty list and cost 4.8 us of a 62 us step.  cont_stride = 0 keeps the work list (long horizons:
// the block-parallel tail consumes it; NMPC_TEAM_INPLACE=0; the builds without CONT).  A team's results do not depend on which wave
// continues it (the permutation tests), so the two schedules give the same bits.
template <bool SHARED, bool TRAJ, bool LDSC, bool CONT, class TI>
__device__ __forceinline__ void team_as_kernel(const Consts<double> &c, const Work<double> &w, const Inputs<TI> &in, const Outputs<TI> &out,
                                               const TeamWork<double> &tw, const WorkList &wl, int B, int tpw, double *smem, int lds_stride,
                                               int lstg, int lm_off, int pass_cap, double *tail_ts, int cont_stride, int cont_lstg)
{
    // (pass_cap > 0: long horizons - the attempt is handed to the block-parallel tail after that many passes, tail_ts its state rows)
    TailCtx tcx;
    tcx.cap = pass_cap; tcx.ts = tail_ts;
    tcx.inplace = (CONT && cont_stride > 0) ? 1 : 0;
    const bool gave = team_as<SHARED, TRAJ, LDSC, TI>(c, w, in, out, tw, wl, B, tpw, smem, lds_stride, lstg, lm_off, -2, tcx);
    if constexpr (CONT) {
        if (cont_stride > 0 && __ballot(gave) != 0) {
            __threadfence();                 // the hand-over (pass budget, certificate base, pin codes, warm start) is read back below
            __syncthreads();
            const int team = (threadIdx.x >> 2) & 3;
            const int inst = gave ? (int)blockIdx.x * tpw + team : -1;
            // (inlined on purpose: behind a call the arguments become generic pointers and the HOT path turns to flat loads and 244 B of
            // scratch; inlined, the first attempt issues 3 % more instructions than without the continuation - register parking)
            team_as<SHARED, TRAJ, true, TI, 2>(c, w, in, out, tw, wl, B, 4, smem, cont_stride, cont_lstg, lm_off, inst);
        }
    }
}

#endif  // device

}  // namespace nmpc
