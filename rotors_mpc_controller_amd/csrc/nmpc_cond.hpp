// nmpc_cond.hpp -- [UPSTREAM] HPIPM partial condensing + IPM on the condensed OCP-QP,
// the formulation acados actually hands to HPIPM for this reference
// (ocp.solver_options.qp_solver = 'PARTIAL_CONDENSING_HPIPM', qp_solver_cond_N = min(N, 5):
// controller.py:181,184).
//
// The N-stage QP is folded into N2 blocks: block inputs ubar = (u_k0 .. u_k0+bs-1), block state x_k0,
//   x_{k0+j} = Phi_j x_k0 + Gam_j ubar + c_j,   Abar = Phi_bs, Bbar = Gam_bs, bbar = c_bs,
//   Qbar = sum Phi_j' Q Phi_j, Sbar = sum Gam_j' Q Phi_j, Rbar = blkdiag(R) + sum Gam_j' Q Gam_j, ...
// and the same Mehrotra IPM as nmpc_ipm.hpp runs on it with dense stage Hessians, followed by the
// expansion back to the N-stage trajectory.  The solution is identical to the uncondensed path (U8;
// tested), which is why the fast kernels (nmpc_team.hpp / nmpc_ipm.hpp) do NOT condense: with
// diagonal Hessians and nu = 4 the plain Riccati recursion is cheaper than forming dense 16x16 /
// 16x13 blocks.  This path exists for fidelity with the reference's solver configuration and as an
// independent on-device cross-check; it is one-instance-per-lane with every operand in an SoA
// global-memory workspace (generic block size => runtime loops), i.e. deliberately simple, not fast.
#pragma once

#include "nmpc_ipm.hpp"

namespace nmpc {

template <class T>
struct CondWork {
    T *base;        // [rows][Bp]
    int Bp, N2, BS, NUB;           // blocks, max stages per block, max inputs per block
    // row offsets
    int blk0, blk_stride;          // per-block region
    int oA, oB, ob, oQ, oS, oR, oq, orr, oL, oM, om, oU, oLL, oLU, oUA, oDU, oLO, oHI, oTL, oTU, oBS, oQS, oIN, in_stride;
    int oP, oT, oPA, oPB, oV, oXH; // per-lane scratch
    int rows;
};

// fills the offsets; returns the number of rows
template <class T>
inline int cond_layout(CondWork<T> &cw, int N, int N2)
{
    cw.N2 = N2;
    cw.BS = (N + N2 - 1) / N2;
    cw.NUB = NU * cw.BS;
    const int nub = cw.NUB;
    int o = 0;
    cw.oA = o; o += NX * NX;
    cw.oB = o; o += NX * nub;
    cw.ob = o; o += NX;
    cw.oQ = o; o += NX * NX;
    cw.oS = o; o += nub * NX;
    cw.oR = o; o += nub * nub;
    cw.oq = o; o += NX;
    cw.orr = o; o += nub;
    cw.oL = o; o += nub * nub;
    cw.oM = o; o += nub * NX;
    cw.om = o; o += nub;
    cw.oU = o; o += nub; cw.oLL = o; o += nub; cw.oLU = o; o += nub; cw.oUA = o; o += nub;
    cw.oDU = o; o += nub; cw.oLO = o; o += nub; cw.oHI = o; o += nub;
    cw.oTL = o; o += nub; cw.oTU = o; o += nub;      // slacks of the input bounds: iterates of the interior point (oracle ocpqp_ipm)
    cw.oBS = o; o += NX; cw.oQS = o; o += NX;        // b + B u and q + S'u of the iterate: the Newton system is solved for the input STEP
    cw.oIN = o; cw.in_stride = NX * NX + NX * nub + NX; o += (cw.BS + 1) * cw.in_stride;
    cw.blk0 = 0; cw.blk_stride = o;
    int s = N2 * o;
    cw.oP = s; s += NX * NX;
    cw.oT = s; s += NX * NX;
    cw.oPA = s; s += NX * NX;
    cw.oPB = s; s += NX * nub;
    cw.oV = s; s += 8 * NX + 4 * nub;
    cw.oXH = s; s += (N2 + 1) * NX;
    cw.rows = s;
    return s;
}

#define CWV(off) cw.base[(size_t)(off) * (size_t)cw.Bp + (size_t)lane]

template <class T>
NMPC_HD void lane_cond_ipm(const Consts<T> &c, const Work<T> &w, const CondWork<T> &cw, const Inputs<T> &in, const Outputs<T> &out, int lane)
{
    const int N = c.N, Bp = w.Bp, N2 = cw.N2, nubm = cw.NUB;
    const int base_bs = N / N2, rem = N - N2 * base_bs;
    auto bsz = [&](int ib) { return base_bs + (ib < rem ? 1 : 0); };
    auto k0of = [&](int ib) { return ib * base_bs + (ib < rem ? ib : rem); };
    auto A_at = [&](int k, int i, int j) -> T {
        if (j < 3) return i == j ? T(1) : T(0);
        if (j < 6) return i == j - 3 ? c.dt : (i == j ? T(1) : T(0));
        const int cc = j - 6;
        const T *ABk = w.AB + (size_t)(c.shared ? 0 : k) * AB_ROWS * Bp;
        return i < ad_rows(cc) ? NMPC_LD(ABk, ad_ofs(cc) + i) : T(0);
    };
    auto B_at = [&](int k, int i, int j) -> T {
        const T *ABk = w.AB + (size_t)(c.shared ? 0 : k) * AB_ROWS * Bp;
        return NMPC_LD(ABk, AD_SIZE + i * NU + j);
    };
    auto b_at = [&](int k, int i) -> T { return NMPC_LD(w.bv + (size_t)(c.shared ? 0 : k) * NX * Bp, i); };

    // ---------------- condensing
    for (int ib = 0; ib < N2; ib++) {
        const int o = cw.blk0 + ib * cw.blk_stride, nb = bsz(ib), nub = NU * nb, k0 = k0of(ib);
        auto PHI = [&](int jj, int i, int l) { return o + cw.oIN + jj * cw.in_stride + i * NX + l; };
        auto GAM = [&](int jj, int i, int l) { return o + cw.oIN + jj * cw.in_stride + NX * NX + i * nubm + l; };
        auto CC = [&](int jj, int i) { return o + cw.oIN + jj * cw.in_stride + NX * NX + NX * nubm + i; };
        for (int i = 0; i < NX; i++) {
            for (int l = 0; l < NX; l++) CWV(PHI(0, i, l)) = i == l ? T(1) : T(0);
            for (int l = 0; l < nub; l++) CWV(GAM(0, i, l)) = 0;
            CWV(CC(0, i)) = 0;
        }
        for (int jj = 0; jj < nb; jj++) {
            const int k = k0 + jj;
            for (int i = 0; i < NX; i++) {
                for (int l = 0; l < NX; l++) {
                    T s = 0;
                    for (int t = 0; t < NX; t++) s += A_at(k, i, t) * CWV(PHI(jj, t, l));
                    CWV(PHI(jj + 1, i, l)) = s;
                }
                for (int l = 0; l < nub; l++) {
                    T s = 0;
                    for (int t = 0; t < NX; t++) s += A_at(k, i, t) * CWV(GAM(jj, t, l));
                    if (l >= jj * NU && l < (jj + 1) * NU) s += B_at(k, i, l - jj * NU);
                    CWV(GAM(jj + 1, i, l)) = s;
                }
                T s = b_at(k, i);
                for (int t = 0; t < NX; t++) s += A_at(k, i, t) * CWV(CC(jj, t));
                CWV(CC(jj + 1, i)) = s;
            }
        }
        for (int i = 0; i < NX; i++) {
            for (int l = 0; l < NX; l++) { CWV(o + cw.oA + i * NX + l) = CWV(PHI(nb, i, l)); CWV(o + cw.oQ + i * NX + l) = 0; }
            for (int l = 0; l < nub; l++) CWV(o + cw.oB + i * nubm + l) = CWV(GAM(nb, i, l));
            CWV(o + cw.ob + i) = CWV(CC(nb, i));
            CWV(o + cw.oq + i) = 0;
        }
        for (int i = 0; i < nub; i++) {
            for (int l = 0; l < NX; l++) CWV(o + cw.oS + i * NX + l) = 0;
            for (int l = 0; l < nub; l++) CWV(o + cw.oR + i * nubm + l) = 0;
            CWV(o + cw.orr + i) = 0;
        }
        for (int jj = 0; jj < nb; jj++) {
            const int k = k0 + jj;
            // wv = Q c_j + q_k  (kept in the scratch vector area)
            for (int t = 0; t < NX; t++) CWV(cw.oV + t) = c.Qd[t] * CWV(CC(jj, t)) + NMPC_LD(w.qr, k * QR_ROWS + t);
            for (int i = 0; i < NX; i++) {
                for (int l = 0; l < NX; l++) {
                    T s = 0;
                    for (int t = 0; t < NX; t++) s += CWV(PHI(jj, t, i)) * c.Qd[t] * CWV(PHI(jj, t, l));
                    CWV(o + cw.oQ + i * NX + l) += s;
                }
                T s = 0;
                for (int t = 0; t < NX; t++) s += CWV(PHI(jj, t, i)) * CWV(cw.oV + t);
                CWV(o + cw.oq + i) += s;
            }
            for (int i = 0; i < nub; i++) {
                for (int l = 0; l < NX; l++) {
                    T s = 0;
                    for (int t = 0; t < NX; t++) s += CWV(GAM(jj, t, i)) * c.Qd[t] * CWV(PHI(jj, t, l));
                    CWV(o + cw.oS + i * NX + l) += s;
                }
                for (int l = 0; l < nub; l++) {
                    T s = 0;
                    for (int t = 0; t < NX; t++) s += CWV(GAM(jj, t, i)) * c.Qd[t] * CWV(GAM(jj, t, l));
                    CWV(o + cw.oR + i * nubm + l) += s;
                }
                T s = 0;
                for (int t = 0; t < NX; t++) s += CWV(GAM(jj, t, i)) * CWV(cw.oV + t);
                CWV(o + cw.orr + i) += s;
            }
            for (int l = 0; l < NU; l++) {
                const int i = jj * NU + l;
                const T ul = NMPC_LD(w.ul, k * NU + l);
                CWV(o + cw.oR + i * nubm + i) += c.Rd[l];
                CWV(o + cw.orr + i) += NMPC_LD(w.qr, k * QR_ROWS + NX + l);
                CWV(o + cw.oLO + i) = c.lbu[l] - ul;
                CWV(o + cw.oHI + i) = c.ubu[l] - ul;
            }
        }
    }

    // ---------------- IPM on the condensed QP (dense stage Hessians, nu = 4*bs)
    int nc_i = 0;
    for (int ib = 0; ib < N2; ib++) {
        const int o = cw.blk0 + ib * cw.blk_stride, nub = NU * bsz(ib);
        nc_i += 2 * nub;
        for (int i = 0; i < nub; i++) {
            const T lo = CWV(o + cw.oLO + i), hi = CWV(o + cw.oHI + i);
            T thr = c.thr0;
            if (c.thr0_rel * (hi - lo) > thr) thr = c.thr0_rel * (hi - lo);
            if (hi - lo < T(2) * thr) thr = T(0.5) * (hi - lo);
            T v = 0;
            if (v - lo < thr) v = lo + thr;
            if (hi - v < thr) v = hi - thr;
            CWV(o + cw.oU + i) = v;
            CWV(o + cw.oTL + i) = v - lo;
            CWV(o + cw.oTU + i) = hi - v;
            CWV(o + cw.oLL + i) = c.mu0 / (v - lo);
            CWV(o + cw.oLU + i) = c.mu0 / (hi - v);
        }
    }
    const T nc = T(nc_i);
    T rho = T(1), mu = 0;
    int it = 0, status = 0;
    // vector scratch: h[13] at oV, gx at oV+13, pv at oV+26, gu[nub] at oV+39.., rhs stored in oDU during sweeps
    const int vH = cw.oV, vGX = cw.oV + NX, vPV = cw.oV + 2 * NX, vGU = cw.oV + 8 * NX;

    auto backward = [&](bool factor, bool homog) -> bool {
        // rhs gradient per block is expected in oDU (rhat); overwritten by m
        if (factor) {
            for (int i = 0; i < NX; i++)
                for (int l = 0; l < NX; l++) CWV(cw.oP + i * NX + l) = i == l ? c.QdN[i] : T(0);
        }
        for (int i = 0; i < NX; i++) CWV(vPV + i) = homog ? T(0) : NMPC_LD(w.qr, N * QR_ROWS + i);
        for (int ib = N2 - 1; ib >= 0; ib--) {
            const int o = cw.blk0 + ib * cw.blk_stride, nub = NU * bsz(ib);
            for (int i = 0; i < NX; i++) {
                T s = CWV(vPV + i);
                if (!homog)
                    for (int l = 0; l < NX; l++) s += CWV(cw.oP + i * NX + l) * CWV(o + cw.oBS + l);
                CWV(vH + i) = s;
            }
            if (factor) {
                for (int i = 0; i < NX; i++) {
                    for (int l = 0; l < NX; l++) {
                        T s = 0;
                        for (int t = 0; t < NX; t++) s += CWV(cw.oP + i * NX + t) * CWV(o + cw.oA + t * NX + l);
                        CWV(cw.oPA + i * NX + l) = s;
                    }
                    for (int l = 0; l < nub; l++) {
                        T s = 0;
                        for (int t = 0; t < NX; t++) s += CWV(cw.oP + i * NX + t) * CWV(o + cw.oB + t * nubm + l);
                        CWV(cw.oPB + i * nubm + l) = s;
                    }
                }
                for (int i = 0; i < nub; i++) {
                    const T tl = CWV(o + cw.oTL + i), tu = CWV(o + cw.oTU + i);
                    const T sg = CWV(o + cw.oLL + i) / tl + CWV(o + cw.oLU + i) / tu;
                    for (int l = 0; l < nub; l++) {
                        T s = CWV(o + cw.oR + i * nubm + l) + (i == l ? sg : T(0));
                        for (int t = 0; t < NX; t++) s += CWV(o + cw.oB + t * nubm + i) * CWV(cw.oPB + t * nubm + l);
                        CWV(o + cw.oL + i * nubm + l) = s;
                    }
                    for (int l = 0; l < NX; l++) {
                        T s = CWV(o + cw.oS + i * NX + l);
                        for (int t = 0; t < NX; t++) s += CWV(o + cw.oB + t * nubm + i) * CWV(cw.oPA + t * NX + l);
                        CWV(o + cw.oM + i * NX + l) = s;
                    }
                }
                for (int i = 0; i < NX; i++)
                    for (int l = 0; l < NX; l++) {
                        T s = CWV(o + cw.oQ + i * NX + l);
                        for (int t = 0; t < NX; t++) s += CWV(o + cw.oA + t * NX + i) * CWV(cw.oPA + t * NX + l);
                        CWV(cw.oT + i * NX + l) = s;
                    }
                // in-place lower Cholesky of L (nub x nub), then M <- L^{-1} M
                for (int jj = 0; jj < nub; jj++) {
                    T d = CWV(o + cw.oL + jj * nubm + jj);
                    for (int t = 0; t < jj; t++) { const T v = CWV(o + cw.oL + jj * nubm + t); d -= v * v; }
                    if (!(d > T(0))) return false;
                    d = sqrt(d);
                    CWV(o + cw.oL + jj * nubm + jj) = d;
                    for (int i = jj + 1; i < nub; i++) {
                        T s = CWV(o + cw.oL + i * nubm + jj);
                        for (int t = 0; t < jj; t++) s -= CWV(o + cw.oL + i * nubm + t) * CWV(o + cw.oL + jj * nubm + t);
                        CWV(o + cw.oL + i * nubm + jj) = s / d;
                    }
                }
                for (int i = 0; i < nub; i++)
                    for (int l = 0; l < NX; l++) {
                        T s = CWV(o + cw.oM + i * NX + l);
                        for (int t = 0; t < i; t++) s -= CWV(o + cw.oL + i * nubm + t) * CWV(o + cw.oM + t * NX + l);
                        CWV(o + cw.oM + i * NX + l) = s / CWV(o + cw.oL + i * nubm + i);
                    }
                for (int i = 0; i < NX; i++)
                    for (int l = 0; l < NX; l++) {
                        T s = CWV(cw.oT + i * NX + l);
                        for (int t = 0; t < nub; t++) s -= CWV(o + cw.oM + t * NX + i) * CWV(o + cw.oM + t * NX + l);
                        CWV(cw.oPA + i * NX + l) = s;
                    }
                for (int i = 0; i < NX; i++)
                    for (int l = 0; l < NX; l++)
                        CWV(cw.oP + i * NX + l) = T(0.5) * (CWV(cw.oPA + i * NX + l) + CWV(cw.oPA + l * NX + i));
            }
            // gu = rhat + B'h -> m = L^{-1} gu ; gx = q + A'h ; pv = gx - M'm
            for (int i = 0; i < nub; i++) {
                T s = CWV(o + cw.oDU + i);
                for (int t = 0; t < NX; t++) s += CWV(o + cw.oB + t * nubm + i) * CWV(vH + t);
                for (int t = 0; t < i; t++) s -= CWV(o + cw.oL + i * nubm + t) * CWV(o + cw.om + t);
                CWV(o + cw.om + i) = s / CWV(o + cw.oL + i * nubm + i);
            }
            for (int i = 0; i < NX; i++) {
                T s = homog ? T(0) : CWV(o + cw.oQS + i);
                for (int t = 0; t < NX; t++) s += CWV(o + cw.oA + t * NX + i) * CWV(vH + t);
                CWV(vGX + i) = s;
            }
            for (int i = 0; i < NX; i++) {
                T s = CWV(vGX + i);
                for (int t = 0; t < nub; t++) s -= CWV(o + cw.oM + t * NX + i) * CWV(o + cw.om + t);
                CWV(vPV + i) = s;
            }
        }
        return true;
    };
    // forward: result inputs into `dst` (oUA or oDU) per block, block states into oXH
    auto forward = [&](bool homog, int dst) {
        for (int i = 0; i < NX; i++) CWV(cw.oXH + i) = 0;
        for (int ib = 0; ib < N2; ib++) {
            const int o = cw.blk0 + ib * cw.blk_stride, nub = NU * bsz(ib);
            for (int i = 0; i < nub; i++) {
                T s = CWV(o + cw.om + i);
                for (int l = 0; l < NX; l++) s += CWV(o + cw.oM + i * NX + l) * CWV(cw.oXH + ib * NX + l);
                CWV(vGU + i) = -s;
            }
            for (int i = nub - 1; i >= 0; i--) {
                T s = CWV(vGU + i);
                for (int t = i + 1; t < nub; t++) s -= CWV(o + cw.oL + t * nubm + i) * CWV(vGU + t);
                CWV(vGU + i) = s / CWV(o + cw.oL + i * nubm + i);
            }
            for (int i = 0; i < nub; i++) CWV(o + dst + i) = CWV(vGU + i);
            for (int i = 0; i < NX; i++) {
                T s = homog ? T(0) : CWV(o + cw.oBS + i);
                for (int l = 0; l < NX; l++) s += CWV(o + cw.oA + i * NX + l) * CWV(cw.oXH + ib * NX + l);
                for (int l = 0; l < nub; l++) s += CWV(o + cw.oB + i * nubm + l) * CWV(vGU + l);
                CWV(cw.oXH + (ib + 1) * NX + i) = s;
            }
        }
    };

    // one bound pair of the iterate: carried slacks, bound residuals (zero to rounding), and the affine directions of slacks and multipliers
    // for an affine input step d (oracle ocpqp_ipm: el, eu, dla, dua)
    struct PairC { T u, ll, lu, tl, tu, rl, ru; };
    auto pair_at = [&](int o, int i) {
        PairC p;
        p.u = CWV(o + cw.oU + i); p.ll = CWV(o + cw.oLL + i); p.lu = CWV(o + cw.oLU + i);
        p.tl = CWV(o + cw.oTL + i); p.tu = CWV(o + cw.oTU + i);
        p.rl = 0; p.ru = 0;         // (bound residuals: zero in exact arithmetic, left out as in the tile kernels - nmpc_team.hpp, Pair)
        return p;
    };
    for (;;) {
        mu = 0;
        for (int ib = 0; ib < N2; ib++) {
            const int o = cw.blk0 + ib * cw.blk_stride, nub = NU * bsz(ib);
            for (int i = 0; i < nub; i++)
                mu += CWV(o + cw.oLL + i) * CWV(o + cw.oTL + i) + CWV(o + cw.oLU + i) * CWV(o + cw.oTU + i);
        }
        mu /= nc;
        if (!(mu == mu)) { status = 1; break; }
        if (mu <= c.tol_comp && rho <= c.tol_stat) break;
        if (it >= c.iter_max) { status = 2; break; }
        it++;
        for (int ib = 0; ib < N2; ib++) {   // input-step form of the block (b + B u, q + S'u) and the affine rhs r + R u + residual terms into oDU
            const int o = cw.blk0 + ib * cw.blk_stride, nub = NU * bsz(ib);
            for (int i = 0; i < NX; i++) {
                T sb = CWV(o + cw.ob + i), sq = CWV(o + cw.oq + i);
                for (int l = 0; l < nub; l++) {
                    const T ul = CWV(o + cw.oU + l);
                    sb += CWV(o + cw.oB + i * nubm + l) * ul;
                    sq += CWV(o + cw.oS + l * NX + i) * ul;
                }
                CWV(o + cw.oBS + i) = sb;
                CWV(o + cw.oQS + i) = sq;
            }
            for (int i = 0; i < nub; i++) {
                const PairC p = pair_at(o, i);
                T sr = CWV(o + cw.orr + i);
                for (int l = 0; l < nub; l++) sr += CWV(o + cw.oR + i * nubm + l) * CWV(o + cw.oU + l);
                CWV(o + cw.oDU + i) = sr + p.ll / p.tl * p.rl - p.lu / p.tu * p.ru;
            }
        }
        if (!backward(true, false)) { status = 3; break; }
        forward(false, cw.oUA);              // oUA: the affine STEP of the inputs
        T aaff = T(1);
        for (int ib = 0; ib < N2; ib++) {
            const int o = cw.blk0 + ib * cw.blk_stride, nub = NU * bsz(ib);
            for (int i = 0; i < nub; i++) {
                const PairC p = pair_at(o, i);
                const T d = CWV(o + cw.oUA + i), el = d + p.rl, eu = -d + p.ru;
                const T dla = -p.ll - p.ll / p.tl * el, dua = -p.lu - p.lu / p.tu * eu;
                if (el < T(0) && -p.tl / el < aaff) aaff = -p.tl / el;
                if (eu < T(0) && -p.tu / eu < aaff) aaff = -p.tu / eu;
                if (dla < T(0) && -p.ll / dla < aaff) aaff = -p.ll / dla;
                if (dua < T(0) && -p.lu / dua < aaff) aaff = -p.lu / dua;
            }
        }
        T muaff = 0;
        for (int ib = 0; ib < N2; ib++) {
            const int o = cw.blk0 + ib * cw.blk_stride, nub = NU * bsz(ib);
            for (int i = 0; i < nub; i++) {
                const PairC p = pair_at(o, i);
                const T d = CWV(o + cw.oUA + i), el = d + p.rl, eu = -d + p.ru;
                const T dla = -p.ll - p.ll / p.tl * el, dua = -p.lu - p.lu / p.tu * eu;
                muaff += (p.ll + aaff * dla) * (p.tl + aaff * el) + (p.lu + aaff * dua) * (p.tu + aaff * eu);
            }
        }
        muaff /= nc;
        T sg3 = muaff / mu;
        sg3 = sg3 * sg3 * sg3;
        const T sigmu = sg3 * mu;
        for (int ib = 0; ib < N2; ib++) {   // corrector rhs into oDU
            const int o = cw.blk0 + ib * cw.blk_stride, nub = NU * bsz(ib);
            for (int i = 0; i < nub; i++) {
                const PairC p = pair_at(o, i);
                const T da = CWV(o + cw.oUA + i), el = da + p.rl, eu = -da + p.ru;
                const T dla = -p.ll - p.ll / p.tl * el, dua = -p.lu - p.lu / p.tu * eu;
                const T cl = dla * el, cu = dua * eu;
                CWV(o + cw.oDU + i) = -(sigmu - cl) / p.tl + (sigmu - cu) / p.tu;
            }
        }
        backward(false, true);
        forward(true, cw.oDU);
        T amax = T(1e30);
        for (int ib = 0; ib < N2; ib++) {
            const int o = cw.blk0 + ib * cw.blk_stride, nub = NU * bsz(ib);
            for (int i = 0; i < nub; i++) {
                const PairC p = pair_at(o, i);
                const T da = CWV(o + cw.oUA + i), el = da + p.rl, eu = -da + p.ru;
                const T dla = -p.ll - p.ll / p.tl * el, dua = -p.lu - p.lu / p.tu * eu;
                const T cl = dla * el, cu = dua * eu;
                const T d = da + CWV(o + cw.oDU + i);
                CWV(o + cw.oDU + i) = d;
                const T dtl = d + p.rl, dtu = -d + p.ru;
                const T dl = -(p.ll * p.tl + cl - sigmu) / p.tl - p.ll / p.tl * dtl;
                const T du = -(p.lu * p.tu + cu - sigmu) / p.tu - p.lu / p.tu * dtu;
                if (dtl < T(0) && -p.tl / dtl < amax) amax = -p.tl / dtl;
                if (dtu < T(0) && -p.tu / dtu < amax) amax = -p.tu / dtu;
                if (dl < T(0) && -p.ll / dl < amax) amax = -p.ll / dl;
                if (du < T(0) && -p.lu / du < amax) amax = -p.lu / du;
            }
        }
        T alpha = c.tau * amax;
        if (alpha > T(1)) alpha = T(1);
        if (!(alpha == alpha)) { status = 1; break; }
        if (alpha < T(1e-12)) { status = 3; break; }
        for (int ib = 0; ib < N2; ib++) {
            const int o = cw.blk0 + ib * cw.blk_stride, nub = NU * bsz(ib);
            for (int i = 0; i < nub; i++) {
                const PairC p = pair_at(o, i);
                const T da = CWV(o + cw.oUA + i), el = da + p.rl, eu = -da + p.ru;
                const T dla = -p.ll - p.ll / p.tl * el, dua = -p.lu - p.lu / p.tu * eu;
                const T cl = dla * el, cu = dua * eu;
                const T d = CWV(o + cw.oDU + i);
                const T dtl = d + p.rl, dtu = -d + p.ru;
                const T dl = -(p.ll * p.tl + cl - sigmu) / p.tl - p.ll / p.tl * dtl;
                const T du = -(p.lu * p.tu + cu - sigmu) / p.tu - p.lu / p.tu * dtu;
                CWV(o + cw.oU + i) = p.u + alpha * d;
                CWV(o + cw.oTL + i) = p.tl + alpha * dtl;
                CWV(o + cw.oTU + i) = p.tu + alpha * dtu;
                CWV(o + cw.oLL + i) = p.ll + alpha * dl;
                CWV(o + cw.oLU + i) = p.lu + alpha * du;
            }
        }
        rho *= (T(1) - alpha);
    }

    // ---------------- block-state rollout, expansion to the N-stage trajectory, full SQP step
    bool bad = false;
    const bool upd = (status == 0 || status == 2);
    for (int i = 0; i < NX; i++) CWV(cw.oXH + i) = 0;
    for (int ib = 0; ib < N2; ib++) {
        const int o = cw.blk0 + ib * cw.blk_stride, nb = bsz(ib), nub = NU * nb, k0 = k0of(ib);
        for (int i = 0; i < NX; i++) {
            T s = CWV(o + cw.ob + i);
            for (int l = 0; l < NX; l++) s += CWV(o + cw.oA + i * NX + l) * CWV(cw.oXH + ib * NX + l);
            for (int l = 0; l < nub; l++) s += CWV(o + cw.oB + i * nubm + l) * CWV(o + cw.oU + l);
            CWV(cw.oXH + (ib + 1) * NX + i) = s;
        }
        for (int jj = 0; jj < nb; jj++) {
            const int k = k0 + jj;
            for (int l = 0; l < NU; l++) {
                const T du = CWV(o + cw.oU + jj * NU + l);
                bad |= !(du == du);
                if (upd) NMPC_ST(w.ul, k * NU + l, NMPC_LD(w.ul, k * NU + l) + du);
            }
            if (jj > 0 || ib > 0) {   // delta x of stage k (stage 0 is pinned: delta x_0 = 0)
                for (int i = 0; i < NX; i++) {
                    T s = CWV(o + cw.oIN + jj * cw.in_stride + NX * NX + NX * nubm + i);
                    for (int t = 0; t < NX; t++) s += CWV(o + cw.oIN + jj * cw.in_stride + i * NX + t) * CWV(cw.oXH + ib * NX + t);
                    for (int t = 0; t < nub; t++) s += CWV(o + cw.oIN + jj * cw.in_stride + NX * NX + i * nubm + t) * CWV(o + cw.oU + t);
                    bad |= !(s == s);
                    if (upd) NMPC_ST(w.xl, k * NX + i, NMPC_LD(w.xl, k * NX + i) + s);
                }
            }
        }
    }
    for (int i = 0; i < NX; i++) {
        const T s = CWV(cw.oXH + N2 * NX + i);
        bad |= !(s == s);
        if (upd) NMPC_ST(w.xl, N * NX + i, NMPC_LD(w.xl, N * NX + i) + s);
    }
    if (bad && upd) status = 1;
    int nlp_status = (status == 2) ? 0 : (status == 3 ? 4 : status);
    if (nlp_status == 1 || nlp_status == 4) nlp_status = inputs_not_finite(in, N, lane) ? 1 : 4;     // nmpc_ipm.hpp
    w.iters[lane] = it;
    w.status[lane] = nlp_status;
    if (out.status) out.status[lane] = nlp_status;
    for (int i = 0; i < NU; i++) out.u0[(size_t)lane * NU + i] = nlp_status == 0 ? NMPC_LD(w.ul, i) : T(0);
    const bool failed = nlp_status != 0;   // hand back the cold-start point, as lane_ipm does
    if (out.x_out)
        for (int k = 0; k <= N; k++)
            for (int i = 0; i < NX; i++) out.x_out[((size_t)lane * (N + 1) + k) * NX + i] = NMPC_LD(w.xl, (failed ? 0 : k) * NX + i);
    if (out.u_out)
        for (int k = 0; k < N; k++)
            for (int i = 0; i < NU; i++) out.u_out[((size_t)lane * N + k) * NU + i] = failed ? T(0) : NMPC_LD(w.ul, k * NU + i);
}

}  // namespace nmpc
