// nmpc_stage.hpp -- ONE factor stage of the tile-form Riccati recursion (gfx950 device code), the single source of
//   * the backward sweep of the solver's kernels (nmpc_team_as.hpp: k_team_as / k_team_qp / k_team_qp_list / k_team_tail) and
//   * the block sweeps of the parallel-in-time factorisation (nmpc_block.hpp: k_block_sweep).
// Until round 3 nmpc_block.hpp restated this stage "because the solver's one is a lambda over its pass state", and a 1e-11 test was all
// that kept the two together; a stage factorised by either must produce the SAME BITS (the block-parallel tail continues solves the
// sequential kernels began, and results must not depend on who factorised a stage), which one source gives by construction.
//
// The stage (padded homogeneous form): with Pbar the value behind the stage, Abar = [Aq0 | Aq1] the dense column
// tiles of the transition (b_k and the pinned inputs' values in column 15), Bt the input tiles,
//     H = D + Bm'Pbar Bm = L Dh L' (unit L),   X = Bm'Pbar Abar + rhat e15',   M0 = L^-1 X,   M = Dh^-1 M0,
//     Pbar_k = Qbar + Abar'Pbar Abar - M0'M     (kept exactly symmetric: products above the diagonal, transposes below)
// Variants (compile time): PINS - inputs marked active are pinned at their bounds (they leave B through a mask and enter b);
// IPMV - barrier terms of an interior-point iteration on the input Hessian, the stage solved for the STEP of the inputs (the iterate's
// inputs enter b like pinned values); both - the choice is a run-time flag per team (tail mode);
// LAST - stage 0 of a sweep whose value is not needed (no Riccati update).
// What a caller does around it: operand prefetch, the stores of the factors / gradient rows / checkpoints (through the sink), pass logic.
#pragma once

#include "nmpc_team.hpp"

namespace nmpc {

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)

// per-lane constants of the tile-form sweeps (lane (a,c) of a team = element (a,c) of every 4 x 4 tile)
struct StageLane {
    int ta, tc;
    int natR[4];            // natural state index of row a of tile t (-1: pad)
    double Qdg[4];          // diagonal of the stage Hessian in tile layout
    int iq_col[4], iq_row[4];   // where the lane finds its entries of column / row 15 of Qbar in the gradient buffer (or the zero slot)
    double dt_v, Idt, Ihalf;    // dt; identity tile; half of it
    double Rd_a, lb_a, ub_a;    // input a: Hessian diagonal, bounds
    double lbj, ubj;        // input c: bounds
};

// scalars of one stage (fetched a stage ahead by the caller)
struct StageIn {
    double rk;              // W_r (u_lin - yref_u) of input a, a ROUNDED product (see nmpc_team_as.hpp)
    double q_r;             // W_q (x_lin - yref_x) of natural row rr
    double ul, ulc;         // linearisation input of components a | c
    double pc, pca;         // pin codes of inputs c | a (PINS)
    double u_it, ll_it, lu_it;   // iterate of input a (IPMV): input, multipliers of its two bounds ...
    double tl_it, tu_it;         // ... and their slacks (iterates of their own)
};

struct StageOut {
    double Aq1[4];          // the second column tile of Abar with the pinned inputs' values added to column 15 (what the stage used)
    double M[4];            // Mbar = Dh^-1 L^-1 X as tiles
    double Zt;              // L^-1 as a tile
    double Y;               // L^-T as a tile
    double ra;              // 1 / d_a
    double mask_a;          // free mask of input a
    bool any_pins;          // some input of the wave's teams is pinned at this stage (wave-uniform)
};

// sink of three callables (device lambdas of the caller)
template <class G, class H, class F>
struct StageSink {
    G g; H h; F f;
    __device__ __forceinline__ void grad(int jt, double a) { g(jt, a); }
    __device__ __forceinline__ void hr(double v) { h(v); }
    __device__ __forceinline__ void factors(const StageOut &o) { f(o); }
};
template <class G, class H, class F>
__device__ __forceinline__ StageSink<G, H, F> stage_sink(G g, H h, F f) { return StageSink<G, H, F>{g, h, f}; }

// sink: what the stage hands to its caller while it runs (kept at the points of the instruction stream where the solver's sweep stored)
//   grad(jt, a)  raw B'(Pbar Abar) tile jt (gradient rows of pinned inputs), hr(Hr) raw B'Pbar B;  only called when PINS and the stage has pins
//   factors(o)   M, L^-1, 1/d are final
// BtT: the input tiles TRANSPOSED (lane (j,a) of tile kt: B[4 kt + a][j]; pad rows zero) - the operand of B v, read only when PINS or IPMV
template <bool PINS, bool IPMV, bool LAST, bool TRACK_GM, class Sink>
__device__ __forceinline__ void riccati_factor_stage(const StageLane &L, double *sh, double *sHg, const int r,
                                                     const double (&Aq0)[4], const double (&Aq1b)[4], const double (&Bt)[4], const double (&BtT)[4],
                                                     const StageIn &in, const bool pins_on, const bool ipm_on,
                                                     double (&Pt)[4][4], double &gm, bool &ok, bool &nanp, StageOut &o, Sink &&sink)
{
    using T = double;
    const int ta = L.ta, tc = L.tc;
    T mask_a = T(1), mask_c = T(1), D_a = L.Rd_a, rhat_a = in.rk;
    if (IPMV) {
        // barrier terms of the interior-point iteration, solved for the STEP w of the inputs (u = u_it + w; oracle ocpqp_ipm):
        // D = R + lam_l / t_l + lam_u / t_u, rhat = r + R u_it, and b_k + B_k u_it in column 15 below
        const Pair<T> pr(in.ll_it, in.lu_it, in.tl_it, in.tu_it);
        const T sg = pr.kl + pr.ku;
        const T rh = in.rk + L.Rd_a * in.u_it;
        if (PINS) { D_a = ipm_on ? L.Rd_a + sg : D_a; rhat_a = ipm_on ? rh : rhat_a; }
        else { D_a = L.Rd_a + sg; rhat_a = rh; }
    }
    bool any_pins = false;
    T (&Aq1)[4] = o.Aq1;
    NMPC_UNROLL for (int kt = 0; kt < 4; kt++) Aq1[kt] = Aq1b[kt];
    if (!LAST) sh[r] = in.q_r;                   // natural row rr of the stage gradient
    if (PINS) {
        // pinned inputs leave B (free mask) and enter through b (pinned value); their own row keeps R_jj so that u_j = bound
        const bool pinned_a = pins_on && in.pca != T(0), pinned = pins_on && in.pc != T(0);      // input a | input c of this lane
        const T vpin_a = in.pca < T(0) ? L.lb_a - in.ul : L.ub_a - in.ul;
        const T vpin_c = in.pc < T(0) ? L.lbj - in.ulc : L.ubj - in.ulc;
        mask_a = pinned_a ? T(0) : T(1); mask_c = pinned ? T(0) : T(1);
        if (IPMV) { rhat_a = pinned_a ? -L.Rd_a * vpin_a : rhat_a; }
        else { D_a = L.Rd_a; rhat_a = pinned_a ? -L.Rd_a * vpin_a : in.rk; }
        any_pins = __ballot(pinned) != 0;
        if (any_pins || IPMV) {
            // pinned inputs - and the inputs of an interior-point iterate - enter through b (column 15 of Abar): b += B v as one MFMA per row
            // tile, v_j in lane (j, 3).  (Until round 5 a product + a two-step cross-lane sum per tile: 4 ds_bpermute each, the LDS pipe's
            // latency in front of the W = P Abar products that need the column.)
            (void)vpin_c;
            const T vcol = tc == 3 ? (pinned_a ? vpin_a : T(0)) + ((IPMV && ipm_on) ? in.u_it : T(0)) : T(0);
            NMPC_UNROLL for (int kt = 0; kt < 4; kt++) Aq1[kt] = mfma44(BtT[kt], vcol, Aq1[kt]);
        }
    } else if (IPMV) {
        const T vcol = tc == 3 ? in.u_it : T(0);             // b_k + B_k u_it (column 15 of Abar)
        NMPC_UNROLL for (int kt = 0; kt < 4; kt++) Aq1[kt] = mfma44(BtT[kt], vcol, Aq1[kt]);
    }
    // P B and Hr = B'PB first: the factorisation below depends on nothing else
    T WB[4], Hr = 0;
    NMPC_UNROLL for (int it = 0; it < 4; it++) {
        T aB = 0;
        NMPC_UNROLL for (int kt = 0; kt < 4; kt++) aB = mfma44(Pt[kt][it], Bt[kt], aB);
        WB[it] = aB;
    }
    NMPC_UNROLL for (int kt = 0; kt < 4; kt++) Hr = mfma44(Bt[kt], WB[kt], Hr);
    const T Hrm = PINS ? mask_a * mask_c * Hr : Hr;
    if (TRACK_GM) gm = fmax(gm, fabs(Hrm));      // growth certificate: max |B'PB| as the free inputs see it
    const T HuuD = (ta == tc) ? L.Rd_a : T(0);   // diagonal of Huu without pins
    const T Huu = ((PINS || IPMV) ? ((ta == tc) ? D_a : T(0)) : HuuD) + Hrm;
    if (tc <= ta) sHg[lidx(ta, tc)] = Huu;
    NMPC_WSYNC();
    T Lf[10];
    NMPC_UNROLL for (int i = 0; i < 10; i++) Lf[i] = sHg[i];
    // W = Pbar * [Aq0 | Aq1]  (Pbar symmetric: its tile (kt,it) read transposed is tile (it,kt)).  Tile (3,0) of Abar -
    // d omega+ / d q and the homogeneous row - is identically zero (the body rates do not depend on the attitude):
    // its products are left out here, in the (q,w) x (q,w) block below and in the forward sweep (94 MFMAs per stage)
    T W0[4], W1[4];
    NMPC_UNROLL for (int it = 0; it < 4; it++) {
        T a0 = 0, a1 = 0;
        NMPC_UNROLL for (int kt = 0; kt < 4; kt++) {
            if (kt < 3) a0 = mfma44(Pt[kt][it], Aq0[kt], a0);
            a1 = mfma44(Pt[kt][it], Aq1[kt], a1);
        }
        W0[it] = a0; W1[it] = a1;
    }
    T PA[4][4];
    NMPC_UNROLL for (int kt = 0; kt < 4; kt++) {
        PA[kt][0] = Pt[kt][0];
        PA[kt][1] = L.dt_v * Pt[kt][0] + Pt[kt][1];
        PA[kt][2] = W0[kt];
        PA[kt][3] = W1[kt];
    }
    // X = B'(Pbar Abar) (column 15: B'h)
    T X0raw = 0;
    T X[4];
    NMPC_UNROLL for (int jt = 0; jt < 4; jt++) {
        T a = 0;
        if (jt == 0) a = mfma44(WB[0], L.Idt, T(0));
        else if (jt == 1) a = mfma44(WB[1], L.Idt, T(0)) + L.dt_v * X0raw;
        else { NMPC_UNROLL for (int kt = 0; kt < 4; kt++) a = mfma44(Bt[kt], PA[kt][jt], a); }
        if (jt == 0) X0raw = a;
        X[jt] = PINS ? mask_a * a : a;
        // gradient rows of the pinned inputs for the multiplier check of the forward sweep
        if (PINS) { if (any_pins) sink.grad(jt, a); }
    }
    if (PINS) { if (any_pins) sink.hr(Hr); }
    if (tc == 3) X[3] += rhat_a;                                   // gu = rhat + mask * B'h
    // the (q,w) x (q,w) tiles of Abar'(Pbar Abar) and the (p,v) rows: independent of the factorisation
    T Pn[4][4];
    if (!LAST) {
        T qcol[4], qrow[4];       // column / row 15 of Qbar: the stage gradient, zero elsewhere (read from a zero slot)
        NMPC_UNROLL for (int t = 0; t < 4; t++) { qcol[t] = sh[L.iq_col[t]]; qrow[t] = sh[L.iq_row[t]]; }
        // Pbar_k = Qbar + Abar'(Pbar Abar) - Mbar'Mbar, assembled in the MFMA accumulators: the ten tiles on and above
        // the diagonal only (see the update below)
        NMPC_UNROLL for (int jt = 0; jt < 4; jt++) {
            T a2 = (jt == 2 ? L.Qdg[2] : T(0)) + (jt == 3 ? qcol[2] : T(0));
            T a3 = (jt == 3 ? L.Qdg[3] + qcol[3] + qrow[3] : T(0));
            if (jt >= 2) {
                NMPC_UNROLL for (int kt = 0; kt < 4; kt++) {
                    if (kt < 3) a2 = mfma44(Aq0[kt], PA[kt][jt], a2);
                    if (jt == 3) a3 = mfma44(Aq1[kt], PA[kt][jt], a3);
                }
            }
            Pn[0][jt] = PA[0][jt] + (jt == 0 ? L.Qdg[0] : T(0)) + (jt == 3 ? qcol[0] : T(0));
            Pn[1][jt] = jt >= 1 ? L.dt_v * PA[0][jt] + PA[1][jt] + (jt == 1 ? L.Qdg[1] : T(0)) + (jt == 3 ? qcol[1] : T(0)) : T(0);
            Pn[2][jt] = a2;
            Pn[3][jt] = a3;
        }
    }
    // H_uu = L D L' (unit L), replicated in every lane of the team.  Square-root free on purpose: a pivot costs
    // v_rcp_f64 + two Newton steps (4 FMAs) where the Cholesky form cost v_rsq_f64 + 8, the inverse of a UNIT
    // triangle needs 4 FMAs where the general one needed 16 operations, and FP64 vector instructions are paid
    // in full here - they share the SIMD's double-precision pipe with the MFMAs.
    // c_ij = l_ij d_j are the unscaled column entries.
    T r0, r1, r2, r3, l10, l20, l30, l21, l31, l32;
    {
        const T h00 = Lf[lidx(0, 0)], h10 = Lf[lidx(1, 0)], h11 = Lf[lidx(1, 1)], h20 = Lf[lidx(2, 0)], h21 = Lf[lidx(2, 1)];
        const T h22 = Lf[lidx(2, 2)], h30 = Lf[lidx(3, 0)], h31 = Lf[lidx(3, 1)], h32 = Lf[lidx(3, 2)], h33 = Lf[lidx(3, 3)];
        auto pivot = [&](T d) -> T {
            const bool big = !(fabs(d) <= PIVOT_MAX);       // NaN, inf, or beyond what can be squared (nmpc_team.hpp)
            const bool pos = d > T(0) && !big;
            // a NaN / out-of-range pivot counts as NaN DATA only when it is the first pivot of the sweep to fail: the sweep runs on after a
            // pivot that is not positive (the wave's other teams need it), and what it computes for this team from then on is garbage that
            // may well overflow to inf - inf (seeds 431 / 3043 of the fuzz: a warm start from a diverged trajectory - the oracle, which
            // stops at the first failed pivot, reports a QP failure there, and so must this)
            nanp |= ok && big;
            ok &= pos;
            return fast_rcp(pos ? d : T(1));
        };
        r0 = pivot(h00);
        l10 = h10 * r0; l20 = h20 * r0; l30 = h30 * r0;
        r1 = pivot(h11 - l10 * h10);
        const T c21 = h21 - l20 * h10, c31 = h31 - l30 * h10;
        l21 = c21 * r1; l31 = c31 * r1;
        r2 = pivot(h22 - l20 * h20 - l21 * c21);
        const T c32 = h32 - l30 * h20 - l31 * c21;
        l32 = c32 * r2;
        r3 = pivot(h33 - l30 * h30 - l31 * c31 - l32 * c32);
    }
    // Y = L^-T as a tile: lane (a,c) holds (L^-1)[c][a].  The inverse of the unit triangle in closed form and
    // one select by the lane's (c,a): straight-line code - a forward substitution on the unit vector e_a, as
    // the general kernel does it, compiles to lane-divergent branches that cut the stage's scheduling region
    T Y;
    {
        // the six entries below the diagonal of L^-1, NEGATED (p_ij = -(L^-1)_ij: the same roundings as the signed form - a negation is
        // exact -, but the lane's entry is selected first and its sign flipped once, where the signed form cost six sign flips)
        const T p10 = l10, p21 = l21, p32 = l32;
        const T p20 = l20 - l21 * p10;
        const T p31 = l31 - l32 * p21;
        const T p30 = l30 - l31 * p10 - l32 * p20;
        const int e = tc * 4 + ta;           // (row c, column a) of L^-1
        Y = (ta == tc) ? T(-1) : T(0);
        Y = e == 4 ? p10 : Y;  Y = e == 8 ? p20 : Y;  Y = e == 9 ? p21 : Y;
        Y = e == 12 ? p30 : Y; Y = e == 13 ? p31 : Y; Y = e == 14 ? p32 : Y;
        Y = -Y;
    }
    // M0 = L^-1 X, M = D^-1 M0 (row a of the tile by 1 / d_a): the feedback is u = -L^-T (M xbar), the Riccati
    // update subtracts M0' M
    const T ra = ta == 0 ? r0 : (ta == 1 ? r1 : (ta == 2 ? r2 : r3));
    T M0[4];
    NMPC_UNROLL for (int jt = 0; jt < 4; jt++) {
        M0[jt] = mfma44(Y, X[jt], T(0));
        o.M[jt] = ra * M0[jt];
    }
    o.Zt = mfma44(Y, L.Idt, T(0));                                // Y' = L^-1 as a tile
    o.Y = Y; o.ra = ra; o.mask_a = mask_a; o.any_pins = any_pins;
    sink.factors(o);
    if (!LAST) {
        // Pbar is kept EXACTLY symmetric: products for the tiles on and above the diagonal, the diagonal tiles
        // averaged with their transposes, the tiles below as transposes (X' I transposes a tile).  Computed
        // independently, tile (i,j) and tile (j,i) differ by rounding, and that antisymmetric part is not
        // contracted by the recursion: it grows by rho(A)^2 per stage - harmless for the reference's vehicle
        // (rho = 1.04), a NaN after ~30 stages where the discretised open loop is violently unstable (dt = 0.1 with
        // one integrator step and a small inertia: rho = 2; found by tools/dev/fuzz_parity.py).  The
        // lane kernel and the oracle carry one triangle of P only.
        NMPC_UNROLL for (int it = 0; it < 4; it++) {
            NMPC_UNROLL for (int jt = it; jt < 4; jt++) Pt[it][jt] = mfma44_na(M0[it], o.M[jt], Pn[it][jt]);      // - M0'M (negated operand: exact)
        }
        // (X + X') / 2 as X'(I / 2) + X / 2: the halves are exact, so this is the rounded sum the two-step form gave, one addition less
        NMPC_UNROLL for (int it = 0; it < 4; it++) Pt[it][it] = mfma44(Pt[it][it], L.Ihalf, T(0.5) * Pt[it][it]);
        NMPC_UNROLL for (int it = 1; it < 4; it++) {
            NMPC_UNROLL for (int jt = 0; jt < it; jt++) Pt[it][jt] = mfma44(Pt[jt][it], L.Idt, T(0));
        }
    }
}

#endif  // device

}  // namespace nmpc
