// nmpc_ipm.hpp -- the two per-lane phases of one SQP real-time iteration:
//
//   lane_prepare   [UPSTREAM ocp_nlp_sqp_rti preparation]  staging of controller.py:414-445
//                  (x0 pin, cold/warm initial trajectory, yref) + linearisation of every
//                  shooting interval + LINEAR_LS Gauss-Newton gradients.
//   lane_ipm       [UPSTREAM HPIPM]  Mehrotra predictor-corrector interior point method on the
//                  OCP-QP with the Riccati factorisation of nmpc_lane.hpp, then the full SQP
//                  step x += dx, u += du (controller.py:447, outputs read at :452-460).
//
// Same algorithm, same update rules and the same constants as oracle/nmpc_oracle.c
// (ocpqp_ipm), restated for one-instance-per-lane execution:
//   * feasible start in the inputs; the slacks t_l, t_u of the input bounds are ITERATES of their own (HPIPM's form:
//     t <- t + alpha dt, never re-formed as u - lo), states are implied by the affine dynamics -> the residuals that
//     matter are stationarity (shrinks by 1-alpha per step, tracked as rho) and complementarity (mu);
//   * each Newton system is an LQ problem - absolute in the states, solved for the STEP of the inputs (u = u_it + w: the
//     direction of a nearly active input must resolve its 1e-14 slack, which a target ua with d = ua - u cannot) - by one
//     backward and one forward Riccati sweep; the corrector re-uses the factorisation through a
//     homogeneous sweep for the change of the input gradient;
//   * the primal-dual update of an iteration is applied lazily inside the next backward
//     sweep, so the small per-stage vectors cross memory once per sweep.
#pragma once

#include "nmpc_lane.hpp"

namespace nmpc {

// user-side arrays (row-major / AoS exactly as the C ABI takes them)
template <class T>
struct Inputs {
    const T *x0;      // [B][13]
    const T *yref;    // [B][N][17] or [N][17]
    const T *yref_e;  // [B][13]    or [13]
    const T *x_init;  // [B][N+1][13] or null
    const T *u_init;  // [B][N][4]    or null
    int yref_bcast;
};

// How a FAILED solve is classed (every kernel's last step; oracle orc_sqp_rti): NMPC_NAN_DETECTED (1) exactly when one of the instance's own
// INPUTS - x0, yref, yref_e, the linearisation trajectory - is not finite, NMPC_QP_FAILURE (4) otherwise.  The class used to follow the
// arithmetic (a NaN pivot or a NaN in the step: 1), and on data that have left the range of the model - a warm start about a diverged
// trajectory, |x| 5e3 .. 5e10, pivots of 1e116 .. 1e269 - that made it a matter of whose sums overflowed or cancelled first: fuzz draws
// 431 and 11856 ended 1 on one side and 4 on the other.  What the caller handed in does not depend on rounding.  (The reference treats
// every non-zero status alike: controller.py:448-450.)
template <class T>
NMPC_HD bool inputs_not_finite(const Inputs<T> &in, int N, int inst)
{
    bool nf = false;
    auto scan = [&](const T *p, int n) { for (int i = 0; i < n; i++) { const double v = (double)p[i]; nf |= !(fabs(v) <= 1.7976931348623157e308); } };
    scan(in.x0 + (size_t)inst * NX, NX);
    scan(in.yref_bcast ? in.yref : in.yref + (size_t)inst * N * NY, N * NY);
    scan(in.yref_bcast ? in.yref_e : in.yref_e + (size_t)inst * NX, NX);
    if (in.x_init && in.u_init) { scan(in.x_init + (size_t)inst * (N + 1) * NX, (N + 1) * NX); scan(in.u_init + (size_t)inst * N * NU, N * NU); }
    return nf;
}

template <class T>
struct Outputs {
    T *u0;            // [B][4]
    T *x_out;         // [B][N+1][13] or null
    T *u_out;         // [B][N][4] or null
    int32_t *status = nullptr;   // [B] caller's status array (device) or null; the workspace copy is always kept
};

template <class T>
NMPC_HD void lane_prepare(const Consts<T> &c, const Work<T> &w, const Inputs<T> &in, int lane)
{
    const int N = c.N, Bp = w.Bp;
    const bool warm = in.x_init != nullptr && in.u_init != nullptr;
    const T *x0 = in.x0 + (size_t)lane * NX;
    const T *yr = in.yref_bcast ? in.yref : in.yref + (size_t)lane * N * NY;
    const T *ye = in.yref_bcast ? in.yref_e : in.yref_e + (size_t)lane * NX;
    const T *xi = warm ? in.x_init + (size_t)lane * (N + 1) * NX : nullptr;
    const T *ui = warm ? in.u_init + (size_t)lane * N * NU : nullptr;

    // trajectory staging (controller.py:414-431) and cost gradients (U4)
    for (int k = 0; k <= N; k++) {
        T xk[NX];
        NMPC_UNROLL for (int i = 0; i < NX; i++) {
            xk[i] = (warm && k > 0) ? xi[(size_t)k * NX + i] : x0[i];   // stage 0 is pinned to x0
            NMPC_ST(w.xl, k * NX + i, xk[i]);
        }
        if (k < N) {
            NMPC_UNROLL for (int i = 0; i < NX; i++)
                NMPC_ST(w.qr, k * QR_ROWS + i, c.Wq[i] * (xk[i] - yr[(size_t)k * NY + i]));
            NMPC_UNROLL for (int i = 0; i < NU; i++) {
                const T uk = warm ? ui[(size_t)k * NU + i] : T(0);
                NMPC_ST(w.ul, k * NU + i, uk);
                NMPC_ST(w.qr, k * QR_ROWS + NX + i, c.Wr[i] * (uk - yr[(size_t)k * NY + NX + i]));
            }
        } else {
            NMPC_UNROLL for (int i = 0; i < NX; i++)
                NMPC_ST(w.qr, N * QR_ROWS + i, c.WqN[i] * (xk[i] - ye[i]));
        }
    }
    // linearisation (U2, U3): one interval if the cold start lets all stages share it
    const int Ns = c.shared ? 1 : N;
    for (int k = 0; k < Ns; k++) {
        T xk[NX], uk[NU], xn[NX], S[11][NX];
        NMPC_UNROLL for (int i = 0; i < NX; i++) xk[i] = NMPC_LD(w.xl, k * NX + i);
        NMPC_UNROLL for (int i = 0; i < NU; i++) uk[i] = NMPC_LD(w.ul, k * NU + i);
        erk_sens(c, xk, uk, xn, S);
        T *ABk = w.AB + (size_t)k * AB_ROWS * Bp;
        NMPC_UNROLL for (int cc = 0; cc < NZ; cc++) {
            NMPC_UNROLL for (int r = 0; r < ad_rows(cc); r++) NMPC_ST(ABk, ad_ofs(cc) + r, S[cc][r]);
        }
        NMPC_UNROLL for (int i = 0; i < NX; i++) {
            NMPC_UNROLL for (int j = 0; j < NU; j++) NMPC_ST(ABk, AD_SIZE + i * NU + j, S[7 + j][i]);
        }
        NMPC_UNROLL for (int i = 0; i < NX; i++)
            NMPC_ST(w.bv, k * NX + i, xn[i] - NMPC_LD(w.xl, (k + 1) * NX + i));
    }
}

template <class T>
NMPC_HD void lane_ipm(const Consts<T> &c, const Work<T> &w, const Inputs<T> &in, const Outputs<T> &out, int lane)
{
    const int N = c.N, Bp = w.Bp;
    const T nc = T(2 * NU) * T(N);
    const size_t abs_ = c.shared ? 0 : (size_t)AB_ROWS * Bp;
    const size_t bs_ = c.shared ? 0 : (size_t)NX * Bp;
    const size_t lms_ = (size_t)LM_ROWS * Bp, ivs_ = (size_t)IV_ROWS * Bp;

    // ---- initial point: inputs pushed inside the box, perfectly centred multipliers
    for (int k = 0; k < N; k++) {
        T *ivk = w.iv + k * ivs_;
        NMPC_UNROLL for (int i = 0; i < NU; i++) {
            const T ul = NMPC_LD(w.ul, k * NU + i);
            const T lo = c.lbu[i] - ul, hi = c.ubu[i] - ul;
            T thr = c.thr0;
            if (c.thr0_rel * (hi - lo) > thr) thr = c.thr0_rel * (hi - lo);
            if (hi - lo < T(2) * thr) thr = T(0.5) * (hi - lo);
            T v = 0;
            if (v - lo < thr) v = lo + thr;
            if (hi - v < thr) v = hi - thr;
            NMPC_ST(ivk, i, v);
            NMPC_ST(ivk, IV_TL + i, v - lo);
            NMPC_ST(ivk, IV_TU + i, hi - v);
            NMPC_ST(ivk, 4 + i, c.mu0 / (v - lo));
            NMPC_ST(ivk, 8 + i, c.mu0 / (hi - v));
        }
    }
    NMPC_PROF_BEGIN
    T mu = c.mu0, rho = T(1), alpha = 0, sigmu = 0;
    int it = 0, status = 0;
    bool pending = false;
    // one bound pair of the iterate as a sweep sees it (oracle ocpqp_ipm): carried slacks
    struct PairL { T u, ll, lu, tl, tu, rl, ru; };
    auto pair_at = [&](const T *ivk, int k, int i) {
        PairL p;
        const T ul = NMPC_LD(w.ul, k * NU + i);
        p.u = NMPC_LD(ivk, i); p.ll = NMPC_LD(ivk, 4 + i); p.lu = NMPC_LD(ivk, 8 + i);
        p.tl = NMPC_LD(ivk, IV_TL + i); p.tu = NMPC_LD(ivk, IV_TU + i);
        // residuals of the bound equations (HPIPM's res_d): identically zero in exact arithmetic with this feasible start; the oracle can feed
        // them back (orc_config.qp_bound_res) and shows that nothing changes - the kernels leave them out (nmpc_team.hpp, Pair)
        (void)ul;
        p.rl = 0; p.ru = 0;
        return p;
    };
    // full direction of the pair for the affine step da and the final step d (slots 12.., 16.. of the iterate row)
    auto step_of = [&](const PairL &p, T da, T d, T &dtl, T &dtu, T &dl, T &du) {
        const T el = da + p.rl, eu = -da + p.ru;
        const T dla = -p.ll - p.ll / p.tl * el, dua = -p.lu - p.lu / p.tu * eu;
        const T cl = dla * el, cu = dua * eu;
        dtl = d + p.rl; dtu = -d + p.ru;
        dl = -(p.ll * p.tl + cl - sigmu) / p.tl - p.ll / p.tl * dtl;
        du = -(p.lu * p.tu + cu - sigmu) / p.tu - p.lu / p.tu * dtu;
    };

    for (;;) {
        if (!(mu == mu)) { status = 1; break; }
        if (mu <= c.tol_comp && rho <= c.tol_stat) break;
        if (it >= c.iter_max) { status = 2; break; }
        it++;
        // ---- sweep A: (lazy update of the previous step) + backward factorisation,
        //      affine right-hand side
        T P[91], pv[NX];
        NMPC_UNROLL for (int i = 0; i < 91; i++) P[i] = 0;
        NMPC_UNROLL for (int i = 0; i < NX; i++) { P[sidx(i, i)] = c.QdN[i]; pv[i] = NMPC_LD(w.qr, N * QR_ROWS + i); }
        bool ok = true;
        T musum = 0;
        for (int k = N - 1; k >= 0; k--) {
            T *ivk = w.iv + k * ivs_;
            T D[NU], rh[NU], ush[NU];
            NMPC_UNROLL for (int i = 0; i < NU; i++) {
                PairL p = pair_at(ivk, k, i);
                if (pending) {
                    T dtl, dtu, dl, du;
                    step_of(p, NMPC_LD(ivk, 12 + i), NMPC_LD(ivk, 16 + i), dtl, dtu, dl, du);
                    NMPC_ST(ivk, i, p.u + alpha * NMPC_LD(ivk, 16 + i));
                    NMPC_ST(ivk, IV_TL + i, p.tl + alpha * dtl); NMPC_ST(ivk, IV_TU + i, p.tu + alpha * dtu);
                    NMPC_ST(ivk, 4 + i, p.ll + alpha * dl); NMPC_ST(ivk, 8 + i, p.lu + alpha * du);
                    p = pair_at(ivk, k, i);
                }
                musum += p.ll * p.tl + p.lu * p.tu;
                const T sg = p.ll / p.tl + p.lu / p.tu;
                D[i] = c.Rd[i] + sg;
                ush[i] = p.u;
                rh[i] = NMPC_LD(w.qr, k * QR_ROWS + NX + i) + c.Rd[i] * p.u + p.ll / p.tl * p.rl - p.lu / p.tu * p.ru;
            }
            ok &= ricc_factor_stage(c, P, pv, w.AB + k * abs_, w.bv + k * bs_, w.qr + (size_t)k * QR_ROWS * Bp,
                                    D, rh, w.LM + k * lms_, Bp, lane, k == 0, ush);
        }
        pending = false;
        NMPC_STAMP(0)
        mu = musum / nc;   // exact duality measure of the current iterate
        if (!ok) { status = (mu == mu) ? 4 : 1; break; }
        // ---- sweep B: forward affine solve (the STEP of the inputs), step length and mu of the affine step
        T xh[NX], uh[NU], ush[NU];
        NMPC_UNROLL for (int i = 0; i < NX; i++) xh[i] = 0;
        T aaff = T(1), s2 = 0;
        for (int k = 0; k < N; k++) {
            T *ivk = w.iv + k * ivs_;
            NMPC_UNROLL for (int i = 0; i < NU; i++) ush[i] = NMPC_LD(ivk, i);
            ricc_forward_stage(c, xh, uh, w.AB + k * abs_, w.bv + k * bs_, w.LM + k * lms_, Bp, lane,
                               true, k == 0, k == N - 1, ush);
            NMPC_UNROLL for (int i = 0; i < NU; i++) {
                const PairL p = pair_at(ivk, k, i);
                NMPC_ST(ivk, 12 + i, uh[i]);
                const T el = uh[i] + p.rl, eu = -uh[i] + p.ru;
                const T dla = -p.ll - p.ll / p.tl * el, dua = -p.lu - p.lu / p.tu * eu;
                if (el < T(0) && -p.tl / el < aaff) aaff = -p.tl / el;
                if (eu < T(0) && -p.tu / eu < aaff) aaff = -p.tu / eu;
                if (dla < T(0) && -p.ll / dla < aaff) aaff = -p.ll / dla;
                if (dua < T(0) && -p.lu / dua < aaff) aaff = -p.lu / dua;
                s2 += dla * el + dua * eu;
            }
        }
        NMPC_STAMP(1)
        // sum_i (lam + a dlam)(t + a dt) = (1 - a) sum lam t + a^2 sum dlam dt   (sigma = 0)
        const T muaff = (T(1) - aaff) * mu + aaff * aaff * s2 / nc;
        T sg3 = muaff / mu;
        sg3 = sg3 * sg3 * sg3;
        sigmu = sg3 * mu;
        // ---- sweep D: backward homogeneous solve for the corrector's gradient change
        NMPC_UNROLL for (int i = 0; i < NX; i++) pv[i] = 0;
        for (int k = N - 1; k >= 0; k--) {
            T *ivk = w.iv + k * ivs_;
            T dr[NU];
            NMPC_UNROLL for (int i = 0; i < NU; i++) {
                const PairL p = pair_at(ivk, k, i);
                const T da = NMPC_LD(ivk, 12 + i);
                const T el = da + p.rl, eu = -da + p.ru;
                const T dla = -p.ll - p.ll / p.tl * el, dua = -p.lu - p.lu / p.tu * eu;
                const T cl = dla * el, cu = dua * eu;
                dr[i] = -(sigmu - cl) / p.tl + (sigmu - cu) / p.tu;
            }
            ricc_back_homog_stage(c, pv, w.AB + k * abs_, dr, w.LM + k * lms_, Bp, lane, k == 0);
        }
        NMPC_STAMP(2)
        // ---- sweep E: forward homogeneous solve, final direction, step length
        NMPC_UNROLL for (int i = 0; i < NX; i++) xh[i] = 0;
        T amax = T(1e30);
        for (int k = 0; k < N; k++) {
            T *ivk = w.iv + k * ivs_;
            ricc_forward_stage(c, xh, uh, w.AB + k * abs_, w.bv + k * bs_, w.LM + k * lms_, Bp, lane,
                               false, k == 0, k == N - 1);
            NMPC_UNROLL for (int i = 0; i < NU; i++) {
                const PairL p = pair_at(ivk, k, i);
                const T da = NMPC_LD(ivk, 12 + i), d = da + uh[i];
                NMPC_ST(ivk, 16 + i, d);
                T dtl, dtu, dl, du;
                step_of(p, da, d, dtl, dtu, dl, du);
                if (dtl < T(0) && -p.tl / dtl < amax) amax = -p.tl / dtl;
                if (dtu < T(0) && -p.tu / dtu < amax) amax = -p.tu / dtu;
                if (dl < T(0) && -p.ll / dl < amax) amax = -p.ll / dl;
                if (du < T(0) && -p.lu / du < amax) amax = -p.lu / du;
            }
        }
        NMPC_STAMP(3)
        alpha = c.tau * amax;
        if (alpha > T(1)) alpha = T(1);
        if (!(alpha == alpha)) { status = 1; break; }
        if (alpha < T(1e-12)) { status = 3; break; }
        pending = true;
        rho *= (T(1) - alpha);
        // duality measure after the step, for the termination test only (sweep A recomputes
        // it exactly): apply the update to a running sum
        {
            T ms = 0;
            for (int k = 0; k < N; k++) {
                T *ivk = w.iv + k * ivs_;
                NMPC_UNROLL for (int i = 0; i < NU; i++) {
                    const PairL p = pair_at(ivk, k, i);
                    T dtl, dtu, dl, du;
                    step_of(p, NMPC_LD(ivk, 12 + i), NMPC_LD(ivk, 16 + i), dtl, dtu, dl, du);
                    ms += (p.ll + alpha * dl) * (p.tl + alpha * dtl) + (p.lu + alpha * du) * (p.tu + alpha * dtu);
                }
            }
            mu = ms / nc;
        }
        NMPC_STAMP(4)
    }

    NMPC_STAMP(5)
    // ---- final sweep: pending update of the inputs, state rollout, full SQP step (U1)
    {
        T dx[NX];
        NMPC_UNROLL for (int i = 0; i < NX; i++) dx[i] = 0;
        bool bad = false;
        for (int k = 0; k < N; k++) {
            T *ivk = w.iv + k * ivs_;
            T du[NU];
            NMPC_UNROLL for (int i = 0; i < NU; i++) {
                T u = NMPC_LD(ivk, i);
                if (pending) u += alpha * NMPC_LD(ivk, 16 + i);
                du[i] = u;
                bad |= !(u == u);
            }
            T Ad[AD_SIZE], Bm[NX][NU], y[NX];
            const T *ABk = w.AB + k * abs_;
            load_ad(ABk, Bp, lane, Ad);
            load_b(ABk, Bp, lane, Bm);
            NMPC_UNROLL for (int i = 0; i < NX; i++) y[i] = NMPC_LD(w.bv + k * bs_, i);
            a_mul_add(c, Ad, Bm, dx, du, y);
            NMPC_UNROLL for (int i = 0; i < NX; i++) { dx[i] = y[i]; bad |= !(y[i] == y[i]); }
            if (status == 0 || status == 2) {
                NMPC_UNROLL for (int i = 0; i < NU; i++) NMPC_ST(w.ul, k * NU + i, NMPC_LD(w.ul, k * NU + i) + du[i]);
                NMPC_UNROLL for (int i = 0; i < NX; i++)
                    NMPC_ST(w.xl, (k + 1) * NX + i, NMPC_LD(w.xl, (k + 1) * NX + i) + dx[i]);
            }
        }
        if (bad && (status == 0 || status == 2)) status = 1;
    }
    NMPC_STAMP(6)
    NMPC_PROF_END(w)
    // acados RTI tolerates a QP that stopped at its iteration cap (U10)
    int nlp_status = (status == 2) ? 0 : (status == 3 ? 4 : status);
    if (nlp_status == 1 || nlp_status == 4) nlp_status = inputs_not_finite(in, N, lane) ? 1 : 4;     // the class of a failure: see inputs_not_finite
    w.iters[lane] = it;
    w.status[lane] = nlp_status;
    if (out.status) out.status[lane] = nlp_status;
    NMPC_UNROLL for (int i = 0; i < NU; i++)
        out.u0[(size_t)lane * NU + i] = nlp_status == 0 ? NMPC_LD(w.ul, i) : T(0);   // controller.py:448-452
    // a failed instance hands back the cold-start point (x_k = x0, u_k = 0): fed back as a warm start it
    // restarts cold, as the reference does after a failure (controller.py:425-431, 448-450)
    const bool failed = nlp_status != 0;
    if (out.x_out) {
        for (int k = 0; k <= N; k++) {
            NMPC_UNROLL for (int i = 0; i < NX; i++)
                out.x_out[((size_t)lane * (N + 1) + k) * NX + i] = NMPC_LD(w.xl, (failed ? 0 : k) * NX + i);
        }
    }
    if (out.u_out) {
        for (int k = 0; k < N; k++) {
            NMPC_UNROLL for (int i = 0; i < NU; i++)
                out.u_out[((size_t)lane * N + k) * NU + i] = failed ? T(0) : NMPC_LD(w.ul, k * NU + i);
        }
    }
}

}  // namespace nmpc
