// nmpc_team.hpp -- QP phase with 16 lanes per MPC instance ("team" mapping), gfx950 device code.
//
// Why: with one instance per lane a batch of 4096 is only 64 waves on a chip with 1024 SIMDs and
// each wave streams ~40 KB of private workspace per instance through HBM/L2 -- measured
// memory-latency bound (profiles/).  Here a wave holds 4 instances; lane r of a team
// owns ROW r of the 13x13 Riccati matrix and of the stage matrices, the replicated operands
// (B, the 7 dense columns of A, P*A, M) are exchanged through LDS, and one wave per workgroup
// makes every __syncthreads() a single-wave barrier.  B = 4096 -> 1024 waves = one per SIMD.
//
// Same algorithm and constants as lane_ipm() in nmpc_ipm.hpp and as the oracle; only the
// distribution of the arithmetic over lanes differs ([UPSTREAM] HPIPM Riccati IPM, reached by
// the reference through AcadosOcpSolver.solve(), controller.py:447).
//
// Per-instance scratch in HBM is "array of structures" (a team reads contiguous runs):
//   tLM [inst][stage][72] : M column-major [13][4] (52) | L packed, diagonal inverted (10) | m (4) | pad
//   tIV [inst][stage][20] : u | lam_l | lam_u | u_aff | du   (4 each)
// Inputs of the stage matrices come from the SoA workspace written by k_prepare.
#pragma once

#include "nmpc_lane.hpp"

namespace nmpc {

constexpr int TEAM = 16;            // lanes per instance (one DPP row)
constexpr int TEAMS_PER_WAVE = 4;
constexpr int TLM_ROWS = 72;
// LDS carve per team, in elements of T
constexpr int L_AD = 0;             // [16][8]   rows of the dense A columns
constexpr int L_B = L_AD + 128;     // [16][4]
constexpr int L_BV = L_B + 64;      // [16]
constexpr int L_PB = L_BV + 16;     // [16][4]
constexpr int L_H = L_PB + 64;      // [16]
constexpr int L_PA = L_H + 16;      // [16][14]  rows of P*A
constexpr int L_HG = L_PA + 224;    // [16]      Huu (10) | gu (4)
constexpr int L_MC = L_HG + 16;     // [16][4]   columns of M
constexpr int L_D = L_MC + 64;      // [4] D | [4] rhat
constexpr int L_Y = L_D + 8;        // [2][16][4] partial products M[:,c]*x_c, double buffered
constexpr int L_XH = L_Y + 128;     // [2][16]
constexpr int L_DR = L_XH + 32;     // [2][4]
constexpr int L_RED = L_DR + 8;     // [32] small reductions
// 808 elements: as bytes (6464 B FP64 / 3232 B FP32) the team stride is 64 B resp. 160 B past a
// multiple of the 256-B LDS bank row, so the four teams of a wave - which issue the same relative
// address at the same time - fall on different banks.  (A stride of 800 doubles = 25 bank rows
// made every broadcast read a 4-way conflict: SQ_LDS_BANK_CONFLICT was 20 % of the wave cycles.)
constexpr int TEAM_LDS = L_RED + 40;

template <class T>
struct TeamWork {
    T *tLM;
    T *tIV;
};

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)

// Exchange point inside a stage.  The workgroup is ONE wave and a wave's LDS instructions execute
// in issue order, so a ds_read issued after a ds_write of another lane already sees the data: no
// s_barrier and - unlike __syncthreads() - no s_waitcnt vmcnt(0) that would drain the global
// stores/prefetches in flight.  The builtin only pins the instruction order for the compiler.
// Global data handed between lanes (L, m written by lane 0) crosses real __syncthreads() at the
// sweep boundaries.
#ifndef NMPC_TEAM_FULL_SYNC
#define NMPC_WSYNC() __builtin_amdgcn_wave_barrier()
#else
#define NMPC_WSYNC() __syncthreads()
#endif

template <class T>
__device__ __forceinline__ T sel4(const T *v, int j)
{
    return j == 0 ? v[0] : (j == 1 ? v[1] : (j == 2 ? v[2] : v[3]));
}

template <class T>
__device__ __forceinline__ void team_ipm(const Consts<T> &c, const Work<T> &w, const Outputs<T> &out,
                                         const TeamWork<T> &tw, int B, T *smem)
{
    const int tid = threadIdx.x, team = tid >> 4, r = tid & 15;
    const int rr = r < NX ? r : NX - 1;   // row used for loads; rows 13..15 shadow row 12 and never store
    const int j = r & 3;                  // input component handled by lanes r < 4 (others shadow)
    const bool rowl = r < NX, cmpl = r < NU;
    int inst = blockIdx.x * (blockDim.x >> 4) + team;   // 1, 2 or 4 teams per wave (launch decides)
    const bool valid = inst < B;
    if (!valid) inst = B - 1;             // idle teams shadow the last instance and never store
    const int N = c.N, Bp = w.Bp, lane = inst;
    const T nc = T(2 * NU) * T(N);
    const size_t abs_ = c.shared ? 0 : (size_t)AB_ROWS * Bp, bs_ = c.shared ? 0 : (size_t)NX * Bp;
    T *S = smem + team * TEAM_LDS;
    T *sAd = S + L_AD, *sB = S + L_B, *sbv = S + L_BV, *sPB = S + L_PB, *sh = S + L_H, *sPA = S + L_PA;
    T *sHg = S + L_HG, *sMc = S + L_MC, *sD = S + L_D, *sY = S + L_Y, *sXh = S + L_XH, *sDr = S + L_DR;
    T *sRed = S + L_RED;
    T *tLM = tw.tLM + (size_t)inst * N * TLM_ROWS, *tIV = tw.tIV + (size_t)inst * N * IV_ROWS;

    // row r of the stage matrices and column r of A, in registers
    T Adrow[NZ], Brow[NU], b_r = 0, Acol[NX];
    auto load_stage = [&](int k) {
        const T *ABk = w.AB + k * abs_;
        NMPC_UNROLL for (int cc = 0; cc < NZ; cc++) {
            const T v = NMPC_LD(ABk, ad_ofs(cc) + (rr < ad_rows(cc) ? rr : 0));
            Adrow[cc] = rr < ad_rows(cc) ? v : T(0);
            sAd[r * 8 + cc] = Adrow[cc];
        }
        sAd[r * 8 + 7] = 0;
        NMPC_UNROLL for (int i = 0; i < NU; i++) {
            Brow[i] = NMPC_LD(ABk, AD_SIZE + rr * NU + i);
            sB[r * 4 + i] = Brow[i];
        }
        b_r = NMPC_LD(w.bv + k * bs_, rr);
        sbv[r] = b_r;
        NMPC_WSYNC();
        const int cz = rr >= 6 ? rr - 6 : 0;
        NMPC_UNROLL for (int l = 0; l < NX; l++) {
            const T zc = sAd[l * 8 + cz];
            const T e = (l == rr) ? T(1) : T(0);
            const T ev = (l == rr - 3) ? c.dt : T(0);
            Acol[l] = rr >= 6 ? zc : (rr >= 3 ? e + ev : e);
        }
    };
    if (c.shared) load_stage(0);

    const T lbj = sel4(c.lbu, j), ubj = sel4(c.ubu, j), Rdj = sel4(c.Rd, j);
    // ---- initial point
    for (int k = 0; k < N; k++) {
        const T ul = NMPC_LD(w.ul, k * NU + j);
        const T lo = lbj - ul, hi = ubj - ul;
        T thr = c.thr0;
        if (c.thr0_rel * (hi - lo) > thr) thr = c.thr0_rel * (hi - lo);
        if (hi - lo < T(2) * thr) thr = T(0.5) * (hi - lo);
        T v = 0;
        if (v - lo < thr) v = lo + thr;
        if (hi - v < thr) v = hi - thr;
        if (cmpl && valid) {
            T *ivk = tIV + k * IV_ROWS;
            ivk[j] = v;
            ivk[4 + j] = c.mu0 / (v - lo);
            ivk[8 + j] = c.mu0 / (hi - v);
        }
    }
    __syncthreads();
    NMPC_PROF_BEGIN
    T mu = c.mu0, rho = T(1), alpha = 0, sigmu = 0;
    int it = 0, status = 0;
    bool pending = false, done = false;
    const T Qdr = [&] { T v = 0; NMPC_UNROLL for (int i = 0; i < NX; i++) v = (i == rr) ? c.Qd[i] : v; return v; }();
    const T QdNr = [&] { T v = 0; NMPC_UNROLL for (int i = 0; i < NX; i++) v = (i == rr) ? c.QdN[i] : v; return v; }();

    for (;;) {
        // per-team termination test; the wave keeps sweeping until all four teams are done
        if (!done) {
            if (!(mu == mu)) { status = 1; done = true; }
            else if (mu <= c.tol_comp && rho <= c.tol_stat) done = true;
            else if (it >= c.iter_max) { status = 2; done = true; }
        }
        if (__ballot(!done) == 0) break;
        const bool act = !done;           // frozen teams keep computing but never store
        const bool st_ok = act && valid;
        if (act) it++;

        // ================= sweep A: lazy update + backward factorisation, affine rhs
        T Prow[NX], pv;
        NMPC_UNROLL for (int cc = 0; cc < NX; cc++) Prow[cc] = (cc == rr) ? QdNr : T(0);
        pv = NMPC_LD(w.qr, N * QR_ROWS + rr);
        bool ok = true;
        T musum = 0;
        for (int k = N - 1; k >= 0; k--) {
            if (!c.shared) load_stage(k);
            T *ivk = tIV + k * IV_ROWS, *lmk = tLM + k * TLM_ROWS;
            {
                const T ul = NMPC_LD(w.ul, k * NU + j);
                const T lo = lbj - ul, hi = ubj - ul;
                T u = ivk[j], ll = ivk[4 + j], lu = ivk[8 + j];
                if (pending) {
                    const T tl = u - lo, tu = hi - u;
                    const T da = ivk[12 + j] - u, d = ivk[16 + j];
                    const T dla = -ll - ll / tl * da, dua = -lu + lu / tu * da;
                    const T cl = dla * da, cu = -dua * da;
                    const T dl = -(ll * tl + cl - sigmu) / tl - ll / tl * d;
                    const T du = -(lu * tu + cu - sigmu) / tu + lu / tu * d;
                    u += alpha * d; ll += alpha * dl; lu += alpha * du;
                    if (cmpl && st_ok) { ivk[j] = u; ivk[4 + j] = ll; ivk[8 + j] = lu; }
                }
                const T tl = u - lo, tu = hi - u;
                musum += ll * tl + lu * tu;
                const T sg = ll / tl + lu / tu;
                if (cmpl) { sD[j] = Rdj + sg; sD[4 + j] = NMPC_LD(w.qr, k * QR_ROWS + NX + j) - sg * u; }
            }
            const T q_r = NMPC_LD(w.qr, k * QR_ROWS + rr);
            // P1: row r of P*B, P*b + p, P*A
            T PBrow[NU], h = pv, PArow[NX];
            NMPC_UNROLL for (int i = 0; i < NU; i++) PBrow[i] = 0;
            NMPC_UNROLL for (int cc = 0; cc < NZ; cc++) PArow[6 + cc] = 0;
            NMPC_UNROLL for (int l = 0; l < NX; l++) {
                const T pl = Prow[l];
                NMPC_UNROLL for (int i = 0; i < NU; i++) PBrow[i] += pl * sB[l * 4 + i];
                h += pl * sbv[l];
                NMPC_UNROLL for (int cc = 0; cc < NZ; cc++) {
                    if (l < ad_rows(cc)) PArow[6 + cc] += pl * sAd[l * 8 + cc];
                }
            }
            NMPC_UNROLL for (int i = 0; i < 3; i++) { PArow[i] = Prow[i]; PArow[3 + i] = c.dt * Prow[i] + Prow[3 + i]; }
            NMPC_UNROLL for (int i = 0; i < NU; i++) sPB[r * 4 + i] = PBrow[i];
            sh[r] = h;
            NMPC_UNROLL for (int cc = 0; cc < NX; cc++) sPA[r * 14 + cc] = PArow[cc];
            NMPC_WSYNC();
            // P2: lanes 0..9 one entry of Huu = D + B'PB each, lanes 10..13 one entry of gu = rhat + B'h
            {
                const int e = r < 14 ? r : 13;
                const int ei = e < 10 ? (e >= 6 ? 3 : (e >= 3 ? 2 : (e >= 1 ? 1 : 0))) : e - 10;
                const int ej = e < 10 ? e - ei * (ei + 1) / 2 : 0;
                T a = e < 10 ? (ei == ej ? sD[ei] : T(0)) : sD[4 + ei];
                NMPC_UNROLL for (int l = 0; l < NX; l++) a += sB[l * 4 + ei] * (e < 10 ? sPB[l * 4 + ej] : sh[l]);
                sHg[r] = a;
            }
            NMPC_WSYNC();
            // P3: Cholesky (replicated), column r of M, p_k
            T Lf[10], mv[NU], Mcol[NU];
            NMPC_UNROLL for (int i = 0; i < 10; i++) Lf[i] = sHg[i];
            NMPC_UNROLL for (int i = 0; i < NU; i++) mv[i] = sHg[10 + i];
            NMPC_UNROLL for (int jj = 0; jj < NU; jj++) {
                T d = Lf[lidx(jj, jj)];
                NMPC_UNROLL for (int l = 0; l < jj; l++) d -= Lf[lidx(jj, l)] * Lf[lidx(jj, l)];
                if (!(d > T(0))) { ok = false; d = T(1); }
                const T rd = nmpc_rsqrt(d);
                Lf[lidx(jj, jj)] = rd;
                NMPC_UNROLL for (int i = jj + 1; i < NU; i++) {
                    T a = Lf[lidx(i, jj)];
                    NMPC_UNROLL for (int l = 0; l < jj; l++) a -= Lf[lidx(i, l)] * Lf[lidx(jj, l)];
                    Lf[lidx(i, jj)] = a * rd;
                }
            }
            l_solve(Lf, mv);
            if (r == 0 && st_ok) {
                NMPC_UNROLL for (int i = 0; i < 10; i++) lmk[52 + i] = Lf[i];
                NMPC_UNROLL for (int i = 0; i < NU; i++) lmk[62 + i] = mv[i];
            }
            T gx = q_r;
            NMPC_UNROLL for (int i = 0; i < NU; i++) Mcol[i] = 0;
            NMPC_UNROLL for (int l = 0; l < NX; l++) {
                NMPC_UNROLL for (int i = 0; i < NU; i++) Mcol[i] += Acol[l] * sPB[l * 4 + i];
                gx += Acol[l] * sh[l];
            }
            l_solve(Lf, Mcol);
            NMPC_UNROLL for (int i = 0; i < NU; i++) sMc[r * 4 + i] = Mcol[i];
            if (rowl && st_ok) {
                NMPC_UNROLL for (int i = 0; i < NU; i++) lmk[rr * 4 + i] = Mcol[i];
            }
            T pvn = gx;
            NMPC_UNROLL for (int i = 0; i < NU; i++) pvn -= Mcol[i] * mv[i];
            NMPC_WSYNC();
            // P4: row r of P_k = Q + A'(PA) - M'M
            if (k > 0) {
                NMPC_UNROLL for (int cc = 0; cc < NX; cc++) {
                    T a = (cc == rr) ? Qdr : T(0);
                    NMPC_UNROLL for (int l = 0; l < NX; l++) a += Acol[l] * sPA[l * 14 + cc];
                    NMPC_UNROLL for (int i = 0; i < NU; i++) a -= Mcol[i] * sMc[cc * 4 + i];
                    Prow[cc] = a;
                }
                pv = pvn;
            }
            NMPC_WSYNC();
        }
        NMPC_STAMP(0)
        if (cmpl) sRed[j] = musum;
        __syncthreads();
        const T mu_now = (sRed[0] + sRed[1] + sRed[2] + sRed[3]) / nc;
        if (act) {
            pending = false;
            mu = mu_now;
            if (!ok) { status = (mu == mu) ? 4 : 1; done = true; }
        }
        const bool act2 = act && !done, st_ok2 = act2 && valid;

        // ================= sweep B: forward affine solve
        T xh = 0, aaff = T(1), s2 = 0;
        int p = 0;
        for (int k = 0; k < N; k++) {
            if (!c.shared) load_stage(k);
            T *ivk = tIV + k * IV_ROWS, *lmk = tLM + k * TLM_ROWS;
            NMPC_UNROLL for (int i = 0; i < NU; i++) sY[p * 64 + r * 4 + i] = lmk[rr * 4 + i] * xh;
            sXh[p * 16 + r] = xh;
            NMPC_WSYNC();
            T uh[NU], Lf[10];
            NMPC_UNROLL for (int i = 0; i < 10; i++) Lf[i] = lmk[52 + i];
            NMPC_UNROLL for (int i = 0; i < NU; i++) {
                T a = lmk[62 + i];
                NMPC_UNROLL for (int cc = 0; cc < NX; cc++) a += sY[p * 64 + cc * 4 + i];
                uh[i] = -a;
            }
            lt_solve(Lf, uh);
            {
                const T ul = NMPC_LD(w.ul, k * NU + j);
                const T lo = lbj - ul, hi = ubj - ul;
                const T u = ivk[j], ll = ivk[4 + j], lu = ivk[8 + j];
                const T uj = sel4(uh, j);
                if (cmpl && st_ok2) ivk[12 + j] = uj;
                const T tl = u - lo, tu = hi - u, d = uj - u;
                const T dla = -ll - ll / tl * d, dua = -lu + lu / tu * d;
                if (d < T(0) && -tl / d < aaff) aaff = -tl / d;
                if (d > T(0) && tu / d < aaff) aaff = tu / d;
                if (dla < T(0) && -ll / dla < aaff) aaff = -ll / dla;
                if (dua < T(0) && -lu / dua < aaff) aaff = -lu / dua;
                s2 += dla * d - dua * d;
            }
            if (k < N - 1) {
                T a = b_r;
                a += (rr < 3) ? xh + c.dt * sXh[p * 16 + rr + 3] : (rr < 6 ? xh : T(0));
                NMPC_UNROLL for (int cc = 0; cc < NZ; cc++) a += Adrow[cc] * sXh[p * 16 + 6 + cc];
                NMPC_UNROLL for (int i = 0; i < NU; i++) a += Brow[i] * uh[i];
                xh = a;
            }
            p ^= 1;
        }
        NMPC_STAMP(1)
        if (cmpl) { sRed[4 + j] = aaff; sRed[8 + j] = s2; }
        __syncthreads();
        aaff = fmin(fmin(sRed[4], sRed[5]), fmin(sRed[6], sRed[7]));
        s2 = sRed[8] + sRed[9] + sRed[10] + sRed[11];
        T sigmu_new;
        {
            const T muaff = (T(1) - aaff) * mu + aaff * aaff * s2 / nc;
            T sg3 = muaff / mu;
            sg3 = sg3 * sg3 * sg3;
            sigmu_new = sg3 * mu;
        }
        if (act2) sigmu = sigmu_new;

        // ================= sweep D: backward homogeneous solve
        pv = 0;
        p = 0;
        for (int k = N - 1; k >= 0; k--) {
            if (!c.shared) load_stage(k);
            T *ivk = tIV + k * IV_ROWS, *lmk = tLM + k * TLM_ROWS;
            {
                const T ul = NMPC_LD(w.ul, k * NU + j);
                const T lo = lbj - ul, hi = ubj - ul;
                const T u = ivk[j], ll = ivk[4 + j], lu = ivk[8 + j];
                const T tl = u - lo, tu = hi - u, da = ivk[12 + j] - u;
                const T dla = -ll - ll / tl * da, dua = -lu + lu / tu * da;
                const T cl = dla * da, cu = -dua * da;
                if (cmpl) sDr[p * 4 + j] = -(sigmu - cl) / tl + (sigmu - cu) / tu;
            }
            sXh[p * 16 + r] = pv;
            NMPC_WSYNC();
            T mv[NU], Lf[10];
            NMPC_UNROLL for (int i = 0; i < 10; i++) Lf[i] = lmk[52 + i];
            NMPC_UNROLL for (int i = 0; i < NU; i++) {
                T a = sDr[p * 4 + i];
                NMPC_UNROLL for (int l = 0; l < NX; l++) a += sB[l * 4 + i] * sXh[p * 16 + l];
                mv[i] = a;
            }
            l_solve(Lf, mv);
            if (r == 0 && st_ok2) {
                NMPC_UNROLL for (int i = 0; i < NU; i++) lmk[62 + i] = mv[i];
            }
            if (k > 0) {
                T a = 0;
                NMPC_UNROLL for (int l = 0; l < NX; l++) a += Acol[l] * sXh[p * 16 + l];
                NMPC_UNROLL for (int i = 0; i < NU; i++) a -= lmk[rr * 4 + i] * mv[i];
                pv = a;
            }
            p ^= 1;
        }
        NMPC_STAMP(2)
        __syncthreads();   // m of every stage (written by lane 0) visible to the team

        // ================= sweep E: forward homogeneous solve, final direction
        xh = 0;
        p = 0;
        T amax = T(1e30);
        for (int k = 0; k < N; k++) {
            if (!c.shared) load_stage(k);
            T *ivk = tIV + k * IV_ROWS, *lmk = tLM + k * TLM_ROWS;
            NMPC_UNROLL for (int i = 0; i < NU; i++) sY[p * 64 + r * 4 + i] = lmk[rr * 4 + i] * xh;
            sXh[p * 16 + r] = xh;
            NMPC_WSYNC();
            T uh[NU], Lf[10];
            NMPC_UNROLL for (int i = 0; i < 10; i++) Lf[i] = lmk[52 + i];
            NMPC_UNROLL for (int i = 0; i < NU; i++) {
                T a = lmk[62 + i];
                NMPC_UNROLL for (int cc = 0; cc < NX; cc++) a += sY[p * 64 + cc * 4 + i];
                uh[i] = -a;
            }
            lt_solve(Lf, uh);
            {
                const T ul = NMPC_LD(w.ul, k * NU + j);
                const T lo = lbj - ul, hi = ubj - ul;
                const T u = ivk[j], ll = ivk[4 + j], lu = ivk[8 + j];
                const T tl = u - lo, tu = hi - u, da = ivk[12 + j] - u;
                const T dla = -ll - ll / tl * da, dua = -lu + lu / tu * da;
                const T cl = dla * da, cu = -dua * da;
                const T d = da + sel4(uh, j);
                if (cmpl && st_ok2) ivk[16 + j] = d;
                const T dl = -(ll * tl + cl - sigmu) / tl - ll / tl * d;
                const T du = -(lu * tu + cu - sigmu) / tu + lu / tu * d;
                if (d < T(0) && -tl / d < amax) amax = -tl / d;
                if (d > T(0) && tu / d < amax) amax = tu / d;
                if (dl < T(0) && -ll / dl < amax) amax = -ll / dl;
                if (du < T(0) && -lu / du < amax) amax = -lu / du;
            }
            if (k < N - 1) {
                T a = 0;
                a += (rr < 3) ? xh + c.dt * sXh[p * 16 + rr + 3] : (rr < 6 ? xh : T(0));
                NMPC_UNROLL for (int cc = 0; cc < NZ; cc++) a += Adrow[cc] * sXh[p * 16 + 6 + cc];
                NMPC_UNROLL for (int i = 0; i < NU; i++) a += Brow[i] * uh[i];
                xh = a;
            }
            p ^= 1;
        }
        NMPC_STAMP(3)
        if (cmpl) sRed[12 + j] = amax;
        __syncthreads();
        amax = fmin(fmin(sRed[12], sRed[13]), fmin(sRed[14], sRed[15]));
        T alpha_new = c.tau * amax;
        if (alpha_new > T(1)) alpha_new = T(1);
        // ================= duality measure after the step (termination test only)
        T ms = 0;
        for (int k = 0; k < N; k++) {
            T *ivk = tIV + k * IV_ROWS;
            const T ul = NMPC_LD(w.ul, k * NU + j);
            const T lo = lbj - ul, hi = ubj - ul;
            const T u = ivk[j], ll = ivk[4 + j], lu = ivk[8 + j];
            const T tl = u - lo, tu = hi - u;
            const T da = ivk[12 + j] - u, d = ivk[16 + j];
            const T dla = -ll - ll / tl * da, dua = -lu + lu / tu * da;
            const T cl = dla * da, cu = -dua * da;
            const T dl = -(ll * tl + cl - sigmu) / tl - ll / tl * d;
            const T du = -(lu * tu + cu - sigmu) / tu + lu / tu * d;
            ms += (ll + alpha_new * dl) * (tl + alpha_new * d) + (lu + alpha_new * du) * (tu - alpha_new * d);
        }
        NMPC_STAMP(4)
        if (cmpl) sRed[16 + j] = ms;
        __syncthreads();
        ms = sRed[16] + sRed[17] + sRed[18] + sRed[19];
        if (act2) {
            alpha = alpha_new;
            if (!(alpha == alpha)) { status = 1; done = true; }
            else if (alpha < T(1e-12)) { status = 3; done = true; }
            else {
                pending = true;
                rho *= (T(1) - alpha);
                mu = ms / nc;
            }
        }
        __syncthreads();   // sRed is reused by the next iteration
    }

    NMPC_STAMP(5)
    // ---- final sweep: pending update of the inputs, state rollout, full SQP step (U1)
    {
        T dx = 0;
        bool bad = false;
        int p = 0;
        const bool upd = (status == 0 || status == 2);
        for (int k = 0; k < N; k++) {
            if (!c.shared) load_stage(k);
            T *ivk = tIV + k * IV_ROWS;
            T u = ivk[j];
            if (pending) u += alpha * ivk[16 + j];
            if (cmpl) sDr[p * 4 + j] = u;
            sXh[p * 16 + r] = dx;
            NMPC_WSYNC();
            T du[NU];
            NMPC_UNROLL for (int i = 0; i < NU; i++) { du[i] = sDr[p * 4 + i]; bad |= !(du[i] == du[i]); }
            T a = b_r;
            a += (rr < 3) ? dx + c.dt * sXh[p * 16 + rr + 3] : (rr < 6 ? dx : T(0));
            NMPC_UNROLL for (int cc = 0; cc < NZ; cc++) a += Adrow[cc] * sXh[p * 16 + 6 + cc];
            NMPC_UNROLL for (int i = 0; i < NU; i++) a += Brow[i] * du[i];
            dx = a;
            if (upd && valid) {
                if (cmpl) NMPC_ST(w.ul, k * NU + j, NMPC_LD(w.ul, k * NU + j) + u);
                if (rowl) NMPC_ST(w.xl, (k + 1) * NX + rr, NMPC_LD(w.xl, (k + 1) * NX + rr) + dx);
            }
            p ^= 1;
        }
        // NaN anywhere in the step poisons the instance: reduce the flag over the team
        sXh[r] = (dx == dx && !bad) ? T(0) : T(1);
        __syncthreads();
        T nb = 0;
        NMPC_UNROLL for (int l = 0; l < NX; l++) nb += sXh[l];
        if (nb > T(0) && upd) status = 1;
    }
    NMPC_STAMP(6)
    NMPC_PROF_END(w)
    const int nlp_status = (status == 2) ? 0 : (status == 3 ? 4 : status);
    if (valid) {
        if (r == 0) { w.iters[inst] = it; w.status[inst] = nlp_status; }
        if (cmpl) out.u0[(size_t)inst * NU + j] = nlp_status == 0 ? NMPC_LD(w.ul, j) : T(0);   // controller.py:448-452
        if (out.x_out && rowl) {
            for (int k = 0; k <= N; k++) out.x_out[((size_t)inst * (N + 1) + k) * NX + rr] = NMPC_LD(w.xl, k * NX + rr);
        }
        if (out.u_out && cmpl) {
            for (int k = 0; k < N; k++) out.u_out[((size_t)inst * N + k) * NU + j] = NMPC_LD(w.ul, k * NU + j);
        }
    }
}

#endif  // device

}  // namespace nmpc
