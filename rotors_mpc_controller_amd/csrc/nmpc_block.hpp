// nmpc_block.hpp -- parallel-in-time Riccati factorisation of one active-set pass (gfx950 device code).
//
// What it is for: at N = 600 (cfg/rotors_mpc.cfg:9 allows it; controller.py:184 then asks HPIPM for partial condensing into
// qp_solver_cond_N = 5 blocks of 120 stages) one Riccati factorisation is 600 dependent stages, and the instances a long-horizon
// solve leaves to the work list (~100 of 1024) keep 25 waves of a 1024-SIMD machine busy for 0.8 ms per pass.  Condensing a block
// into one stage with 480 inputs - what the reference's CPU solver does - would trade that depth for a 480 x 480 Cholesky per
// block and iteration.  On this machine the blocks pay as PARALLELISM instead: the horizon is cut into J blocks, every block is
// swept by its own team at the same time, and only J small boundary updates stay sequential.
//
// The algebra (tools/dev/block_riccati_model.py holds it in numpy).  Stage k of the pinned LQ problem, in the padded homogeneous
// form of nmpc_team_as.hpp (xbar = (x, 1): the stage gradient sits in row / column 15 of Qbar, b_k and the pinned inputs in column 15
// of Abar, the input gradient in column 15 of X):
//     H = D + Bm'P Bm = L Dh L',   X = Bm'P Abar + rhat e15',   M0 = L^-1 X,   M = Dh^-1 M0,   P_k = Qbar + Abar'P Abar - M0'M
// is a linear-fractional map of the value P behind it.  For a block [s, e) the composition of its stage maps is
//     P_s = J + Psi' T Psi,      T = P_e (I + C P_e)^-1,
// where J is the result of the ordinary sweep started from P_e = 0, Psi the product of that sweep's closed-loop transitions
// Abar - Bm Kbar and C = sum_k (Psi_{k+1} Bm) H^-1 (Psi_{k+1} Bm)' their controllability Gramian weighted by H^-1 (the combination
// rule of Sarkka & Garcia-Fernandez, "Temporal parallelization of dynamic programming and linear quadratic control", restated for a
// backward sweep: with C1 = Bm D^-1 Bm' the Woodbury identity turns every (I + C1 J2)^-1 of that rule into the 4 x 4 factorisation
// the stage performs anyway).  So:
//   launch 1  k_block_sweep   blocks 0 .. J-2: zero-terminal sweep carrying (J, Psi', C) - 84 MFMAs per stage on top of the 93 of the
//                             factor stage; block J-1: the ordinary sweep from the terminal cost, factors stored
//   launch 2  k_block_scan    one team per instance walks the J-1 interior boundaries: T through two 16 x 16 block L D L'
//                             factorisations (T = L (Dp^-1 + L'C L)^-1 L' with P_e = L Dp L': symmetric positive definite
//                             operations only, no cancellation; needs P_xx > 0, i.e. positive state weights)
//   launch 3  k_block_sweep   blocks 0 .. J-2: the ordinary sweep from their boundary value, factors stored
// and, for the forward sweep of a pass (tail mode): the scan walks the boundaries forward once more and leaves the state at the start
// of every block, xbar_e = (I + C P_e)^-1 Psi xbar_s, so that the blocks can sweep forward at the same time too (k_team_tail phase 3).
// The factors are the ones the sequential sweep of nmpc_team_as.hpp leaves (Mbar' tiles, L^-1 tile), equal to rounding times the
// conditioning of the boundary update; tests/test_gpu_block.py compares them at N = 600.
//
// The stage itself is nmpc_stage.hpp - the ONE source of the factor stage, shared with sweepA of nmpc_team_as.hpp (until round 3 it was
// restated here): a stage factorised by a block sweep has the bits the solver's own sweep would give it.  Used by
// nmpc_block_factor_device (a building block with its own entry point) and, in tail mode, by every long-horizon solve (N >= 160: the
// block-parallel tail).
#pragma once

#include "nmpc_ipm.hpp"
#include "nmpc_team.hpp"
#include "nmpc_stage.hpp"

namespace nmpc {

constexpr int BLK_FAC_ROWS = 80;     // factors of one stage: Mbar' as 4 tiles x 16 lanes (64) | L^-1 tile (16)
constexpr int BLK_MAT = 256;         // a 16 x 16 matrix as 16 tiles x 16 lanes
constexpr int BLK_GB = 384;          // what the scan keeps of a boundary for its forward pass: 24 tiles x 16 lanes
constexpr int BLK_LDS = 56;          // doubles of LDS per team (gradient row 16 | zero slot | pad | 4 x 4 exchange 16): 192 B past a bank row

struct BlockWork {
    const double *tAB;     // [Bp + 1][N][TAB_ROWS] stage tiles of the per-stage linearisation (left by the last solve)
    const double *tIV;     // [Bp + 1][N][IV_ROWS]  slots 16..19: pin codes of the accepted pass
    double *agg;           // [Bp + 1][J][3][256]   J | Psi' | C of block j (zero-terminal sweep)
    double *bnd;           // [Bp + 1][J + 1][256]  value at the START of block j (bnd[J]: unused)
    double *bchk;          // [Bp + 1][J + 1][256]  start value of block j as its own sweep of launch 3 recomputes it (diagnostic; may be null)
    double *fac;           // [Bp + 1][N][80]       factors
    int Bp, B, J, M;       // M = stages per block
    // tail mode (the long-horizon work list): instances come from the work list, what is factorised is decided
    // by the instance's tail state, factors go straight into the solver's own rows
    const int *list;       // (compacted) work list of this step, *count entries
    const int *count;
    int *reset_count;      // count of the list the NEXT step's instances are appended to: zeroed by the scan of this step (nobody reads it now)
    const double *ts;      // [Bp + 1][TS_ROWS] tail state (nmpc_team.hpp)
    double *tLM;           // [Bp + 1][N][TLM_ROWS]
    double *binfo;         // [Bp + 1][J][2] per block: max |B'PB| of the final sweep | 0 ok, 1 pivot not positive, 2 NaN pivot
    int shared;            // one linearisation for all stages: tiles of stage 0
    // block-parallel forward sweep (tail mode): the scan also leaves the state at the start of every block
    double *gbuf;          // [Bp + 1][J][384]      factors of the boundary behind block j the scan's forward pass needs (or null: no forward pass)
    double *xb;            // [Bp + 1][J][16]       xbar at the start of block j
    const double *frec;    // [Bp + 1][J][FR_ROWS]  records of the last pass's forward sweep (or null): a block whose pin codes did not change
                           //                       keeps its aggregate - the zero-terminal sweep depends on nothing else
    int fwd_in_sweep;      // tail mode: the scan's forward walk (xb) runs as slice y = J - 1 of launch 3, beside the final sweeps it does not
                           // depend on, instead of at the end of the scan kernel (round 5: 40 us of every step's critical path)
};

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)

// H (4 x 4, symmetric, its lower triangle handed round through `ex`) = L Dh L' in every lane of the team: reciprocals of the
// pivots, the tile of L^-T (lane (a,c): (L^-1)[c][a]) and, on request, the tile of L itself.  forced_last: the last pivot is
// replaced by 1 (the constant of a value function).  ok: cleared by a pivot that is not positive.
struct Ldl4 {
    double r0, r1, r2, r3, Y, Lt;
    bool nan;      // the first failing pivot was NaN or beyond PIVOT_MAX
};
template <bool WANT_L>
__device__ __forceinline__ Ldl4 ldl4(double h, double *ex, int ta, int tc, bool forced_last, bool &ok)
{
    using T = double;
    if (tc <= ta) ex[lidx(ta, tc)] = h;
    NMPC_WSYNC();
    T Lf[10];
    NMPC_UNROLL for (int i = 0; i < 10; i++) Lf[i] = ex[i];
    NMPC_WSYNC();
    const T h00 = Lf[lidx(0, 0)], h10 = Lf[lidx(1, 0)], h11 = Lf[lidx(1, 1)], h20 = Lf[lidx(2, 0)], h21 = Lf[lidx(2, 1)];
    const T h22 = Lf[lidx(2, 2)], h30 = Lf[lidx(3, 0)], h31 = Lf[lidx(3, 1)], h32 = Lf[lidx(3, 2)], h33 = Lf[lidx(3, 3)];
    Ldl4 o;
    o.nan = false;
    auto pivot = [&](T d) -> T {
        const bool big = !(fabs(d) <= PIVOT_MAX);
        const bool pos = d > T(0) && !big;
        o.nan |= ok && big;              // (the first failing pivot decides, NaN or out of range: see riccati_factor_stage)
        ok &= pos;
        return fast_rcp(pos ? d : T(1));
    };
    o.r0 = pivot(h00);
    const T l10 = h10 * o.r0, l20 = h20 * o.r0, l30 = h30 * o.r0;
    o.r1 = pivot(h11 - l10 * h10);
    const T c21 = h21 - l20 * h10, c31 = h31 - l30 * h10;
    const T l21 = c21 * o.r1, l31 = c31 * o.r1;
    o.r2 = pivot(h22 - l20 * h20 - l21 * c21);
    const T c32 = h32 - l30 * h20 - l31 * c21;
    const T l32 = c32 * o.r2;
    const T d3 = h33 - l30 * h30 - l31 * c31 - l32 * c32;
    o.r3 = forced_last ? T(1) : pivot(d3);
    {
        const T i10 = -l10, i21 = -l21, i32 = -l32;
        const T i20 = -(l20 + l21 * i10);
        const T i31 = -(l31 + l32 * i21);
        const T i30 = -(l30 + l31 * i10 + l32 * i20);
        const int e = tc * 4 + ta;           // (row c, column a) of L^-1
        T Y = (ta == tc) ? T(1) : T(0);
        Y = e == 4 ? i10 : Y;  Y = e == 8 ? i20 : Y;  Y = e == 9 ? i21 : Y;
        Y = e == 12 ? i30 : Y; Y = e == 13 ? i31 : Y; Y = e == 14 ? i32 : Y;
        o.Y = Y;
    }
    o.Lt = 0;
    if (WANT_L) {
        const int e = ta * 4 + tc;           // (row a, column c) of L
        T L = (ta == tc) ? T(1) : T(0);
        L = e == 4 ? l10 : L;  L = e == 8 ? l20 : L;  L = e == 9 ? l21 : L;
        L = e == 12 ? l30 : L; L = e == 13 ? l31 : L; L = e == 14 ? l32 : L;
        o.Lt = L;
    }
    return o;
}

// One block [s, e) of one instance per team, swept backwards.
//   AGG:  from P_e = 0, carrying Psi' and C; leaves (J, Psi', C) in agg          (launch 1, blocks 0 .. J-2)
//   !AGG: from the terminal cost (e = N) or from bnd[blk + 1]; leaves the factors of its stages and its start value in bnd[blk]
template <bool AGG, class TI, bool TAIL = false>
__device__ __forceinline__ void block_sweep(const Consts<double> &c, const BlockWork &g, const Inputs<TI> &in, int blk, double *smem,
                                            int inst, bool valid)
{
    // inst: an instance whose inputs may be read; valid: this team factorises it (idle teams sweep along in the spare row)
    using T = double;
    const int tid = threadIdx.x, team = (tid >> 2) & 3, r = ((tid >> 4) << 2) | (tid & 3);
    const int ta = r >> 2, tc = r & 3, j = tc;
    const int rr = r < NX ? r : NX - 1;
    const size_t winst = valid ? (size_t)inst : (size_t)g.Bp;
    const bool ipm = TAIL && valid && (int)g.ts[(size_t)inst * TS_ROWS] == TS_IPM;     // barrier terms of an interior-point iteration instead of pins
    const int N = c.N;
    const int s = blk * g.M, e = (s + g.M < N) ? s + g.M : N;
    T *S = smem + team * BLK_LDS;
    T *sh = S, *sHg = S + 24;
    const bool shared = TAIL && g.shared != 0;
    const bool warm = !shared && in.x_init != nullptr && in.u_init != nullptr;
    const TI *x0p = in.x0 + (size_t)inst * NX;
    const TI *yr = in.yref_bcast ? in.yref : in.yref + (size_t)inst * N * NY;
    const TI *ye = in.yref_bcast ? in.yref_e : in.yref_e + (size_t)inst * NX;
    const TI *xi = warm ? in.x_init + (size_t)inst * (N + 1) * NX : x0p;
    const TI *ui = warm ? in.u_init + (size_t)inst * N * NU : x0p;
    const T *tAB = g.tAB + (size_t)inst * N * TAB_ROWS;
    const T *tIV = g.tIV + (size_t)inst * N * IV_ROWS;
    T *fac = TAIL ? nullptr : g.fac + winst * N * BLK_FAC_ROWS;
    T *tLM = TAIL ? g.tLM + winst * N * TLM_ROWS : nullptr;

    int natR[4], natC[4];
    NMPC_UNROLL for (int t = 0; t < 4; t++) { natR[t] = nat_of(t, ta); natC[t] = nat_of(t, tc); }
    T dt_v = c.dt;
    asm volatile("" : "+v"(dt_v));
    const T x0r = (T)x0p[rr];
    const T Wq_r = c.Wq[rr], WqN_r = c.WqN[rr];
    const T Wr_a = c.Wr[ta];
    const T lbj = c.lbu[j], ubj = c.ubu[j];
    const T lb_a = c.lbu[ta], ub_a = c.ubu[ta], Rd_a = c.Rd[ta];
    T Qdg[4];
    NMPC_UNROLL for (int t = 0; t < 4; t++) {
        const T qd = c.Qd[natR[t] >= 0 ? natR[t] : 0];
        Qdg[t] = (ta == tc && natR[t] >= 0) ? qd : T(0);
    }
    constexpr int ZSLOT = 16;
    int iq_col[4], iq_row[4];
    NMPC_UNROLL for (int t = 0; t < 4; t++) {
        iq_col[t] = (tc == 3 && natR[t] >= 0) ? natR[t] : ZSLOT;
        iq_row[t] = (ta == 3 && natC[t] >= 0) ? natC[t] : ZSLOT;
    }
    if (r == 0) sh[ZSLOT] = T(0);
    const T Idt = (ta == tc) ? T(1) : T(0);
    StageLane SL;                                     // per-lane constants of the factor stage (nmpc_stage.hpp)
    SL.ta = ta; SL.tc = tc; SL.dt_v = dt_v; SL.Idt = Idt; SL.Ihalf = (ta == tc) ? T(0.5) : T(0); SL.Rd_a = Rd_a; SL.lb_a = lb_a; SL.ub_a = ub_a; SL.lbj = lbj; SL.ubj = ubj;
    NMPC_UNROLL for (int t = 0; t < 4; t++) { SL.natR[t] = natR[t]; SL.Qdg[t] = Qdg[t]; SL.iq_col[t] = iq_col[t]; SL.iq_row[t] = iq_row[t]; }
    auto xlin = [&](int k) -> T { return (warm && k > 0) ? (T)xi[(size_t)k * NX + rr] : x0r; };
    auto ulin = [&](int k, int comp) -> T { return warm ? (T)ui[(size_t)k * NU + comp] : T(0); };
    // operands of a stage - its 11 stored tiles and 8 to 11 scalars - arrive TWO stages ahead in two alternating register sets: a block
    // sweep runs on a few waves (the tail's lists are short), its stage is shorter than an HBM round trip, and one stage ahead left part of
    // every load exposed (2.5 us per stage against 1.45 us for the same stage in the solver's sweep, whose tiles sit in LDS)
    struct StOps { T pfs[12], pft[4], yx, yu, xl, ul, pc, pca, ulc, u, ll, lu, tl, tu; };
    auto fetch_ops = [&](int kq, StOps &o) {
        const int kc = kq > s ? kq : s;                     // clamped, not skipped (see sweepB of nmpc_team_as.hpp)
        const int k = shared ? 0 : kc;
        const T *a = tAB + (size_t)k * TAB_ROWS + r;
        NMPC_UNROLL for (int t = 0; t < 12; t++) o.pfs[t] = (t == 9) ? T(0) : a[t * 16];      // tile (3,0) is zero and never stored
        {   // the four input tiles once more, transposed (the operand of b += B v: nmpc_stage.hpp)
            const T *at = tAB + (size_t)k * TAB_ROWS + (tc * 4 + ta);
            NMPC_UNROLL for (int kt = 0; kt < 4; kt++) o.pft[kt] = at[(kt * 3 + 2) * 16];
        }
        o.yx = (T)yr[(size_t)kc * NY + rr]; o.yu = (T)yr[(size_t)kc * NY + NX + ta];
        o.xl = xlin(kc); o.ul = ulin(kc, ta);
        o.pc = tIV[kc * IV_ROWS + 16 + j]; o.pca = tIV[kc * IV_ROWS + 16 + ta]; o.ulc = ulin(kc, j);
        o.u = 0; o.ll = 0; o.lu = 0; o.tl = 1; o.tu = 1;                 // the iterate of input a (interior-point iteration)
        if (TAIL) {
            const T *ivn = tIV + kc * IV_ROWS;
            o.u = ivn[ta];
            { const D2 p_ = ld2(ivn + IVP_L + 2 * ta); o.ll = p_.x; o.lu = p_.y; }
            { const D2 p_ = ld2(ivn + IVP_T + 2 * ta); o.tl = p_.x; o.tu = p_.y; }
        }
    };

    T Pt[4][4];
    if (AGG) {
        NMPC_UNROLL for (int it = 0; it < 4; it++) {
            NMPC_UNROLL for (int jt = 0; jt < 4; jt++) Pt[it][jt] = T(0);
        }
    } else if (e == N) {
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
    } else {
        const T *bp = g.bnd + (winst * (g.J + 1) + blk + 1) * BLK_MAT + r;
        NMPC_UNROLL for (int it = 0; it < 4; it++) {
            NMPC_UNROLL for (int jt = 0; jt < 4; jt++) Pt[it][jt] = bp[(it * 4 + jt) * 16];
        }
    }
    // Phi = Psi' (tile (kt,it) of Phi is the transpose of tile (it,kt) of Psi) and C, upper tiles
    T Ph[AGG ? 4 : 1][AGG ? 4 : 1], Cm[AGG ? 4 : 1][AGG ? 4 : 1];
    if constexpr (AGG) {
        NMPC_UNROLL for (int it = 0; it < 4; it++) {
            NMPC_UNROLL for (int jt = 0; jt < 4; jt++) {
                Ph[it][jt] = (it == jt && ta == tc && (natR[it] >= 0 || (it == 3 && ta == 3))) ? T(1) : T(0);
                Cm[it][jt] = T(0);
            }
        }
    }
    bool ok = true;
    const int ks = e - 1;
    StOps oa, ob;
    fetch_ops(ks, oa);
    fetch_ops(ks - 1, ob);
    T gm = 0;
    bool nanp = false;
    auto stage = [&](int k, StOps &o) {
        T Aq0[4], Aq1b[4], Bt[4], BtT[4];
        NMPC_UNROLL for (int kt = 0; kt < 4; kt++) { Aq0[kt] = o.pfs[kt * 3]; Aq1b[kt] = o.pfs[kt * 3 + 1]; Bt[kt] = o.pfs[kt * 3 + 2]; BtT[kt] = o.pft[kt]; }
        StageIn sin;
        sin.ul = o.ul; sin.pc = o.pc; sin.pca = o.pca; sin.ulc = o.ulc; sin.u_it = o.u; sin.ll_it = o.ll; sin.lu_it = o.lu;
        sin.tl_it = o.tl; sin.tu_it = o.tu;
        T rk = Wr_a * (sin.ul - o.yu);
        sin.q_r = Wq_r * (o.xl - o.yx);
        asm volatile("" : "+v"(rk));
        sin.rk = rk;
        fetch_ops(k - 2, o);                                // this set is free again: the stage two below
        // the stage: nmpc_stage.hpp, the source the solver's own backward sweep uses (pins variant; in tail mode the barrier terms of an
        // interior-point iteration instead, chosen per team by its tail state)
        StageOut so;
        auto sink = stage_sink(
            [&](int jt, T a) { if (TAIL && !AGG) tLM[k * TLM_ROWS + TLM_G + jt * 16 + tc * 4 + ta] = a; },      // gradient rows of the pinned inputs
            [&](T hr) { if (TAIL && !AGG) tLM[k * TLM_ROWS + TLM_G + 64 + tc * 4 + ta] = hr; },
            [&](const StageOut &f) {
                if (!AGG) {
                    T *fk = TAIL ? tLM + (size_t)k * TLM_ROWS + TLM_MT : fac + (size_t)k * BLK_FAC_ROWS;
                    NMPC_UNROLL for (int jt = 0; jt < 4; jt++) fk[jt * 16 + tc * 4 + ta] = f.M[jt];        // where a forward sweep reads it transposed
                    fk[64 + r] = f.Zt;                                                                      // Y' = L^-1 as a tile
                    if (TAIL) tLM[(size_t)k * TLM_ROWS + TLM_RINV + ta + (tc == 0 ? 0 : 4)] = f.ra;        // 1 / d_a for the corrector's solves
                }
            });
        riccati_factor_stage<true, TAIL, false, !AGG>(SL, sh, sHg, r, Aq0, Aq1b, Bt, BtT, sin, !ipm, ipm, Pt, gm, ok, nanp, so, sink);
        const T Y = so.Y, ra = so.ra, mask_a = so.mask_a;
        const T (&Aq1)[4] = so.Aq1;
        const T (&M)[4] = so.M;
        (void)Y; (void)ra; (void)mask_a; (void)Aq1; (void)M;
        if constexpr (AGG) {
            // G' = Bm' Phi (rows of pinned inputs masked), N0 = L^-1 G', Nn = Dh^-1 N0:  C += N0' Nn,  Phi <- Abar' Phi - M' N0
            T N0[4], Nn[4];
            NMPC_UNROLL for (int it = 0; it < 4; it++) {
                T gt = 0;
                NMPC_UNROLL for (int kt = 0; kt < 4; kt++) gt = mfma44(Bt[kt], Ph[kt][it], gt);
                N0[it] = mfma44(Y, mask_a * gt, T(0));
                Nn[it] = ra * N0[it];
            }
            NMPC_UNROLL for (int it = 0; it < 4; it++) {
                NMPC_UNROLL for (int jt = it; jt < 4; jt++) Cm[it][jt] = mfma44(N0[it], Nn[jt], Cm[it][jt]);
            }
            T Pq[4][4];
            NMPC_UNROLL for (int it = 0; it < 4; it++) {
                T a2 = 0, a3 = 0;
                NMPC_UNROLL for (int kt = 0; kt < 4; kt++) {
                    if (kt < 3) a2 = mfma44(Aq0[kt], Ph[kt][it], a2);
                    a3 = mfma44(Aq1[kt], Ph[kt][it], a3);
                }
                Pq[0][it] = Ph[0][it];
                Pq[1][it] = dt_v * Ph[0][it] + Ph[1][it];
                Pq[2][it] = a2;
                Pq[3][it] = a3;
            }
            NMPC_UNROLL for (int jt = 0; jt < 4; jt++) {
                NMPC_UNROLL for (int it = 0; it < 4; it++) Ph[jt][it] = mfma44_na(M[jt], N0[it], Pq[jt][it]);
            }
        }
        NMPC_WSYNC();
    };
    for (int k = ks; k >= s; k -= 2) {
        stage(k, oa);
        if (k - 1 >= s) stage(k - 1, ob);
    }
    if constexpr (AGG) {
        NMPC_UNROLL for (int it = 0; it < 4; it++) Cm[it][it] = T(0.5) * (Cm[it][it] + mfma44(Cm[it][it], Idt, T(0)));
        NMPC_UNROLL for (int it = 1; it < 4; it++) {
            NMPC_UNROLL for (int jt = 0; jt < it; jt++) Cm[it][jt] = mfma44(Cm[jt][it], Idt, T(0));
        }
        T *ap = g.agg + (winst * g.J + blk) * 3 * BLK_MAT + r;
        NMPC_UNROLL for (int it = 0; it < 4; it++) {
            NMPC_UNROLL for (int jt = 0; jt < 4; jt++) {
                ap[(it * 4 + jt) * 16] = ok ? Pt[it][jt] : __builtin_nan("");
                ap[BLK_MAT + (it * 4 + jt) * 16] = Ph[it][jt];
                ap[2 * BLK_MAT + (it * 4 + jt) * 16] = Cm[it][jt];
            }
        }
    } else {
        // start value of the block: the last block hands it to the scan; the others (launch 3) must not touch bnd - their
        // neighbours are reading it - and leave theirs in the diagnostic copy, where it can be compared with the scan's
        if (TAIL) {
            // what the pass / iteration needs to know about this block's factorisation
            sh[r] = gm;
            NMPC_WSYNC();
            T gt = sh[0];
            NMPC_UNROLL for (int i = 1; i < 16; i++) gt = fmax(gt, sh[i]);
            const unsigned long long team_mask = 0x000F000F000F000Full << (4 * team);
            const bool t_bad = (__ballot(!ok) & team_mask) != 0, t_nan = (__ballot(nanp) & team_mask) != 0;
            if (r == 0) {
                T *bi = g.binfo + (winst * g.J + blk) * 2;
                bi[0] = gt; bi[1] = t_nan ? T(2) : (t_bad ? T(1) : T(0));
            }
        }
        T *bb = e == N ? g.bnd : g.bchk;
        if (bb) {
            T *bp = bb + (winst * (g.J + 1) + blk) * BLK_MAT + r;
            NMPC_UNROLL for (int it = 0; it < 4; it++) {
                NMPC_UNROLL for (int jt = 0; jt < 4; jt++) bp[(it * 4 + jt) * 16] = ok ? Pt[it][jt] : __builtin_nan("");
            }
        }
    }
}

// 16 x 16 symmetric positive definite A (upper tiles Au[i][k], k >= i; consumed) = L D L' by tiles: per tile column j the
// 4 x 4 factorisation of the diagonal tile, LT[j][i] = L_ij' (i > j), the tiles of the diagonal blocks L_jj and L_jj^-T, 1 / d.
struct Ldl16 {
    double LT[4][4];      // [j][i], i > j: (L_ij)'
    double Ld[4];         // L_jj as a tile
    double Yd[4];         // L_jj^-T as a tile
    double ra[4], rc[4];  // 1 / d of tile column j by the lane's row a | column c
};
__device__ __forceinline__ void ldl16(double (&Au)[4][4], Ldl16 &o, double *ex, int ta, int tc, bool forced_last, bool &ok)
{
    using T = double;
    NMPC_UNROLL for (int jc = 0; jc < 4; jc++) {
        const Ldl4 f = ldl4<true>(Au[jc][jc], ex, ta, tc, forced_last && jc == 3, ok);
        o.Ld[jc] = f.Lt; o.Yd[jc] = f.Y;
        o.ra[jc] = ta == 0 ? f.r0 : (ta == 1 ? f.r1 : (ta == 2 ? f.r2 : f.r3));
        o.rc[jc] = tc == 0 ? f.r0 : (tc == 1 ? f.r1 : (tc == 2 ? f.r2 : f.r3));
        T UT[4];
        NMPC_UNROLL for (int i = 0; i < 4; i++) {
            if (i > jc) {
                UT[i] = mfma44(f.Y, Au[jc][i], T(0));          // L_jj^-1 A_ji = D_j L_ij'
                o.LT[jc][i] = o.ra[jc] * UT[i];
            }
        }
        NMPC_UNROLL for (int i = 0; i < 4; i++) {
            NMPC_UNROLL for (int k = i; k < 4; k++) {
                if (i > jc) Au[i][k] = mfma44(-UT[i], o.LT[jc][k], Au[i][k]);     // A_ik -= L_ij D_j L_kj'
            }
        }
    }
}

__device__ __forceinline__ void block_scan_forward(const BlockWork &g, int inst, bool valid);

// launch 2: the interior boundaries of one instance, last to first:  P_s = J + Psi' T Psi,  T = L (Dp^-1 + L'C L)^-1 L'
__device__ __forceinline__ void block_scan(const BlockWork &g, double *smem, int inst, bool valid)
{
    using T = double;
    const int tid = threadIdx.x, team = (tid >> 2) & 3, r = ((tid >> 4) << 2) | (tid & 3);
    const int ta = r >> 2, tc = r & 3;
    const int rT = tc * 4 + ta;
    const size_t winst = valid ? (size_t)inst : (size_t)g.Bp;
    T *ex = smem + team * BLK_LDS + 24;
    const T Idt = (ta == tc) ? T(1) : T(0);
    T Pe[4][4];
    {
        const T *bp = g.bnd + (winst * (g.J + 1) + g.J - 1) * BLK_MAT + r;
        NMPC_UNROLL for (int it = 0; it < 4; it++) {
            NMPC_UNROLL for (int jt = 0; jt < 4; jt++) Pe[it][jt] = bp[(it * 4 + jt) * 16];
        }
    }
    // the aggregate of a boundary (J | Psi' | C: 42 tiles of 16 doubles from HBM) is fetched ONE BOUNDARY AHEAD: a boundary is a chain of
    // dependent small factorisations with nothing to hide a 2 us load behind, and sixteen of them in sequence are the latency floor of a
    // tail step
    T nCf[4][4], nPs[4][4], nJt[10];
    auto fetch_agg = [&](int b, T (&Cf_)[4][4], T (&Ps_)[4][4], T (&Jt_)[10]) {
        const T *ap_ = g.agg + (winst * g.J + b) * 3 * BLK_MAT;
        NMPC_UNROLL for (int it = 0; it < 4; it++) {
            NMPC_UNROLL for (int jt = 0; jt < 4; jt++) {
                Cf_[it][jt] = ap_[2 * BLK_MAT + (it * 4 + jt) * 16 + r];
                Ps_[it][jt] = ap_[BLK_MAT + (jt * 4 + it) * 16 + rT];          // Psi = Phi' through the transposed lane index
            }
        }
        int q = 0;
        NMPC_UNROLL for (int k = 0; k < 4; k++) {
            NMPC_UNROLL for (int l = k; l < 4; l++) Jt_[q++] = ap_[(k * 4 + l) * 16 + r];
        }
    };
    if (g.J >= 2) fetch_agg(g.J - 2, nCf, nPs, nJt);
    for (int blk = g.J - 2; blk >= 0; blk--) {
        bool ok = true;
        T Cf[4][4], Ps[4][4], Jt[10];
        NMPC_UNROLL for (int it = 0; it < 4; it++) {
            NMPC_UNROLL for (int jt = 0; jt < 4; jt++) { Cf[it][jt] = nCf[it][jt]; Ps[it][jt] = nPs[it][jt]; }
        }
        NMPC_UNROLL for (int q = 0; q < 10; q++) Jt[q] = nJt[q];
        fetch_agg(blk > 0 ? blk - 1 : 0, nCf, nPs, nJt);            // clamped, not skipped
        // P_e with unit pads (positions 3 and 7 of the padded state carry nothing) = L Dp L', the constant's pivot forced to 1
        T Au[4][4];
        NMPC_UNROLL for (int it = 0; it < 4; it++) {
            NMPC_UNROLL for (int jt = it; jt < 4; jt++) Au[it][jt] = Pe[it][jt] + ((it == jt && it < 2 && ta == 3 && tc == 3) ? T(1) : T(0));
        }
        Ldl16 lp;
        ldl16(Au, lp, ex, ta, tc, true, ok);
        // tiles of L: Lm[m][k], m >= k
        T Lm[4][4];
        NMPC_UNROLL for (int k = 0; k < 4; k++) {
            Lm[k][k] = lp.Ld[k];
            NMPC_UNROLL for (int m = k + 1; m < 4; m++) Lm[m][k] = mfma44(lp.LT[k][m], Idt, T(0));
        }
        // E = Dp^-1 + L' C L (upper tiles): F = C L, E = L' F
        T E[4][4];
        {
            T F[4][4];
            NMPC_UNROLL for (int i = 0; i < 4; i++) {
                NMPC_UNROLL for (int k = 0; k < 4; k++) {
                    T a = 0;
                    NMPC_UNROLL for (int m = k; m < 4; m++) a = mfma44(Cf[m][i], Lm[m][k], a);
                    F[i][k] = a;
                }
            }
            NMPC_UNROLL for (int i = 0; i < 4; i++) {
                NMPC_UNROLL for (int k = i; k < 4; k++) {
                    T a = (i == k && ta == tc) ? lp.ra[i] : T(0);
                    // the pivots the factorisation replaced: 1 / d of a pad is 1, of the constant 1 (forced)
                    NMPC_UNROLL for (int m = i; m < 4; m++) a = mfma44(Lm[m][i], F[m][k], a);
                    E[i][k] = a;
                }
            }
            NMPC_UNROLL for (int i = 0; i < 4; i++) E[i][i] = T(0.5) * (E[i][i] + mfma44(E[i][i], Idt, T(0)));
        }
        Ldl16 le;
        ldl16(E, le, ex, ta, tc, false, ok);
        // Z = Le^-1 (L' Psi), rows of tiles by forward substitution; Psi = Phi' is read through the transposed lane index
        T Z[4][4];
        {
            NMPC_UNROLL for (int i = 0; i < 4; i++) {
                NMPC_UNROLL for (int k = 0; k < 4; k++) {
                    T a = 0;
                    NMPC_UNROLL for (int n = i; n < 4; n++) a = mfma44(Lm[n][i], Ps[n][k], a);         // (L' Psi)[i][k]
                    NMPC_UNROLL for (int m = 0; m < 4; m++) {
                        if (m < i) a = mfma44(-le.LT[m][i], Z[m][k], a);                               // - Le_im Z[m][k]
                    }
                    Z[i][k] = mfma44(le.Yd[i], a, T(0));
                }
            }
        }
        if (g.gbuf) {
            // what the forward pass below needs of this boundary: L (10 tiles), Le' (6), Le_ii^-T (4), 1 / de (4 per-lane values)
            T *gp = g.gbuf + (winst * g.J + blk) * BLK_GB + r;
            int q = 0;
            NMPC_UNROLL for (int k = 0; k < 4; k++) {
                NMPC_UNROLL for (int m = k; m < 4; m++) gp[(q++) * 16] = Lm[m][k];
            }
            NMPC_UNROLL for (int k = 0; k < 4; k++) {
                NMPC_UNROLL for (int m = k + 1; m < 4; m++) gp[(q++) * 16] = le.LT[k][m];
            }
            NMPC_UNROLL for (int k = 0; k < 4; k++) gp[(q++) * 16] = le.Yd[k];
            NMPC_UNROLL for (int k = 0; k < 4; k++) gp[(q++) * 16] = le.ra[k];
        }
        // P_s = J + Z' De^-1 Z
        {
            int q = 0;
            NMPC_UNROLL for (int k = 0; k < 4; k++) {
                NMPC_UNROLL for (int l = k; l < 4; l++) {
                    T a = Jt[q++];
                    NMPC_UNROLL for (int i = 0; i < 4; i++) a = mfma44(Z[i][k], le.ra[i] * Z[i][l], a);
                    Pe[k][l] = a;
                }
            }
        }
        NMPC_UNROLL for (int it = 0; it < 4; it++) Pe[it][it] = T(0.5) * (Pe[it][it] + mfma44(Pe[it][it], Idt, T(0)));
        NMPC_UNROLL for (int it = 1; it < 4; it++) {
            NMPC_UNROLL for (int jt = 0; jt < it; jt++) Pe[it][jt] = mfma44(Pe[jt][it], Idt, T(0));
        }
        T *bp = g.bnd + (winst * (g.J + 1) + blk) * BLK_MAT + r;
        NMPC_UNROLL for (int it = 0; it < 4; it++) {
            NMPC_UNROLL for (int jt = 0; jt < 4; jt++) bp[(it * 4 + jt) * 16] = ok ? Pe[it][jt] : __builtin_nan("");
        }
    }
    if (g.gbuf && !g.fwd_in_sweep) {
        __syncthreads();                      // the tiles above were written by other lanes of this team
        block_scan_forward(g, inst, valid);
    }
}

// the scan's forward walk over the boundaries: xbar at the start of every block, from the aggregates and what the backward walk kept of every
// boundary (gbuf).  Depends on the backward walk only - not on the final sweeps - so in tail mode it runs beside them (fwd_in_sweep).
__device__ __forceinline__ void block_scan_forward(const BlockWork &g, int inst, bool valid)
{
    using T = double;
    const int tid = threadIdx.x, r = ((tid >> 4) << 2) | (tid & 3);
    const int ta = r >> 2, tc = r & 3;
    const int rT = tc * 4 + ta;
    const size_t winst = valid ? (size_t)inst : (size_t)g.Bp;
    {
        // forward over the boundaries: xbar at the start of every block (stage 0: the deviation from x0 is zero, xbar = e15).
        //   xbar_e = (I + C P_e)^-1 Psi xbar_s = y - C L Le^-T De^-1 Le^-1 L' y,   y = Psi xbar_s
        // as six matrix-vector products on tiles (a vector sits in column 0 of its four tiles); every operand tile is read from
        // memory in the orientation its product needs (a transposed tile is the same 16 doubles through the swapped lane index)
        T xt[4];
        NMPC_UNROLL for (int t = 0; t < 4; t++) xt[t] = (t == 3 && ta == 3 && tc == 0) ? T(1) : T(0);
        T *xp = g.xb + winst * g.J * 16;
        // tile index of L[m][k] (m >= k), of Le'[k][m] (m > k), of Le_kk^-T, of 1 / de in the boundary's slot
        auto iL = [](int m, int k) { return (k == 0 ? 0 : (k == 1 ? 4 : (k == 2 ? 7 : 9))) + (m - k); };
        auto iE = [](int k, int m) { return 10 + (k == 0 ? 0 : (k == 1 ? 3 : 5)) + (m - k - 1); };
        // the 76 operand tiles of a boundary, every one in the orientation its product needs (a transposed tile is the same 16 doubles
        // through the swapped lane index), fetched ONE BOUNDARY AHEAD: the walk is a chain of six dependent matrix-vector products per
        // boundary with nothing to hide a load behind (2.7 us per boundary when the loads sat inside the chain)
        struct FwOps { T P1[4][4], L2[4][4], E3[4][4], Y3[4], Rd[4], E5[4][4], Y5[4], L6[4][4], C7[4][4]; };
        auto fetch_fw = [&](int bq, FwOps &f) {
            const int b = bq < g.J - 1 ? bq : (g.J >= 2 ? g.J - 2 : 0);          // clamped, not skipped
            const T *ap = g.agg + (winst * g.J + b) * 3 * BLK_MAT;
            const T *gp = g.gbuf + (winst * g.J + b) * BLK_GB;
            NMPC_UNROLL for (int i = 0; i < 4; i++) {
                NMPC_UNROLL for (int k = 0; k < 4; k++) {
                    f.P1[i][k] = ap[BLK_MAT + (k * 4 + i) * 16 + r];             // Psi[i][k]' = Phi[k][i] as stored
                    f.C7[i][k] = ap[2 * BLK_MAT + (k * 4 + i) * 16 + r];         // C[k][i] (C symmetric)
                    if (k >= i) f.L2[i][k] = gp[iL(k, i) * 16 + r];              // L[k][i]
                    if (k < i) f.E3[i][k] = gp[iE(k, i) * 16 + r];               // Le'[k][i]
                    if (k > i) f.E5[i][k] = gp[iE(i, k) * 16 + rT];              // (Le'[i][k])'
                    if (k <= i) f.L6[i][k] = gp[iL(i, k) * 16 + rT];             // L[i][k]'
                }
                f.Y3[i] = gp[(16 + i) * 16 + r];
                f.Y5[i] = gp[(16 + i) * 16 + rT];
                f.Rd[i] = gp[(20 + i) * 16 + r];
            }
        };
        FwOps fn;
        if (g.J >= 2) fetch_fw(0, fn);
        for (int blk = 0; blk < g.J; blk++) {
            if (tc == 0) { NMPC_UNROLL for (int t = 0; t < 4; t++) xp[blk * 16 + t * 4 + ta] = xt[t]; }
            if (blk == g.J - 1) break;
            const FwOps f = fn;
            fetch_fw(blk + 1, fn);
            T y1[4], y2[4], y3[4], y5[4], y6[4];
            NMPC_UNROLL for (int i = 0; i < 4; i++) {             // y1 = Psi x
                T a = 0;
                NMPC_UNROLL for (int k = 0; k < 4; k++) a = mfma44(f.P1[i][k], xt[k], a);
                y1[i] = a;
            }
            NMPC_UNROLL for (int i = 0; i < 4; i++) {             // y2 = L' y1
                T a = 0;
                NMPC_UNROLL for (int n = i; n < 4; n++) a = mfma44(f.L2[i][n], y1[n], a);
                y2[i] = a;
            }
            NMPC_UNROLL for (int i = 0; i < 4; i++) {             // y3 = Le^-1 y2 (forward substitution)
                T a = y2[i];
                NMPC_UNROLL for (int m = 0; m < 4; m++) { if (m < i) a = mfma44_na(f.E3[i][m], y3[m], a); }
                y3[i] = mfma44(f.Y3[i], a, T(0));
            }
            NMPC_UNROLL for (int i = 3; i >= 0; i--) {            // y5 = Le^-T De^-1 y3 (back substitution)
                T a = f.Rd[i] * y3[i];
                NMPC_UNROLL for (int m = 0; m < 4; m++) { if (m > i) a = mfma44_na(f.E5[i][m], y5[m], a); }
                y5[i] = mfma44(f.Y5[i], a, T(0));
            }
            NMPC_UNROLL for (int i = 0; i < 4; i++) {             // y6 = L y5
                T a = 0;
                NMPC_UNROLL for (int m = 0; m <= i; m++) a = mfma44(f.L6[i][m], y5[m], a);
                y6[i] = a;
            }
            NMPC_UNROLL for (int i = 0; i < 4; i++) {             // x_e = y1 - C y6
                T a = y1[i];
                NMPC_UNROLL for (int m = 0; m < 4; m++) a = mfma44_na(f.C7[i][m], y6[m], a);
                xt[i] = a;
            }
        }
    }
}

#endif  // device

}  // namespace nmpc
