// nmpc_lane.hpp -- per-lane arithmetic of the batched SQP-RTI solver (one MPC instance per
// wavefront lane).  Everything here is straight-line, fully unrolled, fixed-size code on
// registers; global memory is touched only through the SoA accessors LD/ST, whose lane-minor
// layout makes every wave access one contiguous 64*sizeof(T) segment.
//
// What it restates (reference file:line; [UPSTREAM] = acados/HPIPM behaviour the reference
// reaches through AcadosOcpSolver.solve(), controller.py:447):
//   model_f / model_jac   controller.py:267-355 (CasADi expression, hand-differentiated)
//   erk_sens              [UPSTREAM] ERK, 2 stages (explicit midpoint) x num_steps with the
//                         forward variational equations (expl_vde_forw), controller.py:183-188
//   ricc_*                [UPSTREAM] HPIPM Riccati factorisation / solves of the OCP-QP KKT system
//   the IPM itself        nmpc_ipm.hpp
//
// Structure exploited (proved in tests/test_oracle_model.py::test_discrete_jacobian_block_structure):
//   A = [ I  dt*I  Apq  Apw ]      only the 7 columns (q, omega) are stored: "Ad", 79 numbers
//       [ 0   I    Avq  Avw ]      (q columns have no omega rows), B is dense 13x4.
//       [ 0   0    Aqq  Aqw ]
//       [ 0   0     0   Aww ]
// The Hessian blocks are diagonal constants (LINEAR_LS with Vx=[I;0], Vu=[0;I],
// controller.py:226-243) so they live in the constant block, not in memory.
#pragma once

#include <math.h>
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define NMPC_HD __host__ __device__ __forceinline__
#define NMPC_UNROLL _Pragma("unroll")
#else
#define NMPC_HD inline
#define NMPC_UNROLL
#endif

namespace nmpc {

constexpr int NX = 13, NU = 4, NY = 17;
constexpr int NZ = 7;                 // stored columns of A: q (4) and omega (3)
constexpr int AD_SIZE = 79;           // 4*10 + 3*13
constexpr int AB_ROWS = AD_SIZE + NX * NU;   // 131 rows per stage in the AB array
constexpr int LM_ROWS = 10 + NU * NX + NU;   // L (10, diagonal stored inverted) + M (52) + m (4)
constexpr int IV_ROWS = 44;           // u, lam_l, lam_u, affine step, step  (4 each) | 4 spare slots: where lanes that carry no input of
                                      // their own store, so that no store of a tile-form sweep is predicated | u, lam_l, lam_u of the warm start an exhausted
                                      // active-set attempt leaves for the interior point (4 each) | t_l, t_u: the slacks of the input bounds, iterates of the
                                      // interior point (4 each; round 5 - HPIPM's form, oracle ocpqp_ipm)
constexpr int IV_TL = 36, IV_TU = 40;
constexpr int QR_ROWS = NX + NU;      // q_k (13), r_k (4)

NMPC_HD constexpr int ad_rows(int c) { return c < 4 ? 10 : 13; }
NMPC_HD constexpr int ad_ofs(int c) { return c < 4 ? 10 * c : 40 + 13 * (c - 4); }
NMPC_HD constexpr int sidx(int i, int j) { return i <= j ? i * NX - i * (i - 1) / 2 + (j - i) : j * NX - j * (j - 1) / 2 + (i - j); }
NMPC_HD constexpr int lidx(int i, int j) { return i * (i + 1) / 2 + j; }   // lower-tri 4x4, j <= i

// constants of one solver, identical for every lane (kernel argument -> scalar registers)
template <class T>
struct Consts {
    int N, steps, iter_max, shared;   // shared: all stages use stage 0's (Ad, B, b)
    T dt, h;
    T Qd[NX], Rd[NU], QdN[NX];        // Hessian diagonals incl. Levenberg-Marquardt
    T Wq[NX], Wr[NU], WqN[NX];        // gradient weights (cost scaling applied)
    T lbu[NU], ubu[NU];
    T inv_mass, gravity, J[3], invJ[3];
    T fuw[3][NU];                     // d omegadot / d u  (rotor geometry / inertia)
    T rx[NU], ry[NU], rz[NU];
    T tol_comp, tol_stat, mu0, tau, thr0, thr0_rel;
    // active-set polish (team kernel only; the lane and condensed kernels are plain IPM)
    int polish, polish_passes, polish_budget, polish_ckpt;
    T polish_mu, kkt_tol;   // kkt_tol: relative acceptance tolerance of the active-set KKT check
    // accuracy certificate of the Riccati factorisations and the exit rules that go with it (nmpc_config.qp_growth_max ...;
    // FP64 tile kernels of nmpc_team_as.hpp; the oracle restates them in ocpqp_ipm / ocpqp_polish)
    T growth_max, acc_comp, acc_stat, tol_step;
    int maxiter_status;     // U10 switch: status of a QP that hits iter_max (0 tolerated, 2 reported)
    int warm_start;         // an attempt that runs out of passes seeds the interior point (nmpc_config.qp_warm_start)
};

constexpr int TAB_ROWS = 192;       // doubles per stage of the team kernels' per-instance stage block tAB

// SoA workspace: row r of an array is the contiguous run [r*Bp, r*Bp + Bp)
template <class T>
struct Work {
    int Bp;
    T *AB;    // [Ns][131][Bp]   Ns = 1 (shared cold start) or N
    T *bv;    // [Ns][13][Bp]
    T *qr;    // [N][17][Bp] then qN [13][Bp]
    T *xl;    // [N+1][13][Bp]  linearisation point in, updated trajectory out
    T *ul;    // [N][4][Bp]
    T *LM;    // [N][66][Bp]
    T *iv;    // [N][20][Bp]
    int32_t *iters;   // [Bp]
    int32_t *status;  // [Bp]
    int32_t *npol;    // [Bp] active-set passes spent (team kernel), or null
    T *tAB;           // [B][Ns][TAB_ROWS] per-instance copy of (Ad rows | B rows | b) for the team kernel, or null
    T *gbase;         // [Bp] growth certificate: max |B'PB| of the first factorisation of the solve (active-set kernel -> work-list launch), or null
    long long *prof;  // [8][Bp] per-sweep time stamps, NMPC_PROFILE builds only (else null)
};

#if defined(NMPC_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define NMPC_PROF_BEGIN long long prof_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long prof_t_ = wall_clock64();
#define NMPC_STAMP(i) { const long long t_ = wall_clock64(); prof_acc_[i] += t_ - prof_t_; prof_t_ = t_; }
#define NMPC_PROF_END(w) if ((w).prof) { for (int i_ = 0; i_ < 8; i_++) (w).prof[(size_t)i_ * (w).Bp + lane] = prof_acc_[i_]; }
#define NMPC_PROF_SINCE(t0) prof_acc_[7] = prof_t_ - (t0);   // slot 7: kernel entry -> first sweep (fused preparation, start point)
#define NMPC_PROF_NOW() wall_clock64()
#else
#define NMPC_PROF_SINCE(t0)
#define NMPC_PROF_NOW() 0ll
#define NMPC_PROF_BEGIN
#define NMPC_STAMP(i)
#define NMPC_PROF_END(w)
#endif

#define NMPC_LD(p, row) ((p)[(size_t)(row) * (size_t)Bp + (size_t)lane])
#define NMPC_ST(p, row, v) ((p)[(size_t)(row) * (size_t)Bp + (size_t)lane] = (v))

// ---------------------------------------------------------------------------------------
// model  (controller.py:267-355)
template <class T>
NMPC_HD void model_f(const Consts<T> &c, const T *x, const T *u, T *f)
{
    const T qw = x[6], qx = x[7], qy = x[8], qz = x[9];
    const T wx = x[10], wy = x[11], wz = x[12];
    const T Tm = (u[0] + u[1] + u[2] + u[3]) * c.inv_mass;
    f[0] = x[3]; f[1] = x[4]; f[2] = x[5];
    f[3] = T(2) * (qx * qz + qw * qy) * Tm;
    f[4] = T(2) * (qy * qz - qw * qx) * Tm;
    f[5] = (T(1) - T(2) * (qx * qx + qy * qy)) * Tm - c.gravity;
    f[6] = T(0.5) * (-qx * wx - qy * wy - qz * wz);
    f[7] = T(0.5) * (qw * wx + qy * wz - qz * wy);
    f[8] = T(0.5) * (qw * wy + qz * wx - qx * wz);
    f[9] = T(0.5) * (qw * wz + qx * wy - qy * wx);
    T tx = 0, ty = 0, tz = 0;
    NMPC_UNROLL for (int i = 0; i < NU; i++) {
        tx += u[i] * c.ry[i];
        ty -= u[i] * c.rx[i];
        tz += u[i] * c.rz[i];
    }
    f[10] = (tx - (c.J[2] - c.J[1]) * wy * wz) * c.invJ[0];
    f[11] = (ty - (c.J[0] - c.J[2]) * wz * wx) * c.invJ[1];
    f[12] = (tz - (c.J[1] - c.J[0]) * wx * wy) * c.invJ[2];
}

template <class T>
struct Jac {
    T Fvq[3][4], Fqq[4][4], Fqw[4][3], Fww[3][3], r3m[3];
};

template <class T>
NMPC_HD void model_jac(const Consts<T> &c, const T *x, const T *u, Jac<T> &J)
{
    const T qw = x[6], qx = x[7], qy = x[8], qz = x[9];
    const T wx = x[10], wy = x[11], wz = x[12];
    const T Tm = (u[0] + u[1] + u[2] + u[3]) * c.inv_mass;
    const T t2 = T(2) * Tm;
    J.Fvq[0][0] = t2 * qy;  J.Fvq[0][1] = t2 * qz;  J.Fvq[0][2] = t2 * qw; J.Fvq[0][3] = t2 * qx;
    J.Fvq[1][0] = -t2 * qx; J.Fvq[1][1] = -t2 * qw; J.Fvq[1][2] = t2 * qz; J.Fvq[1][3] = t2 * qy;
    J.Fvq[2][0] = 0;        J.Fvq[2][1] = -T(2) * t2 * qx; J.Fvq[2][2] = -T(2) * t2 * qy; J.Fvq[2][3] = 0;
    const T hx = T(0.5) * wx, hy = T(0.5) * wy, hz = T(0.5) * wz;
    J.Fqq[0][0] = 0;  J.Fqq[0][1] = -hx; J.Fqq[0][2] = -hy; J.Fqq[0][3] = -hz;
    J.Fqq[1][0] = hx; J.Fqq[1][1] = 0;   J.Fqq[1][2] = hz;  J.Fqq[1][3] = -hy;
    J.Fqq[2][0] = hy; J.Fqq[2][1] = -hz; J.Fqq[2][2] = 0;   J.Fqq[2][3] = hx;
    J.Fqq[3][0] = hz; J.Fqq[3][1] = hy;  J.Fqq[3][2] = -hx; J.Fqq[3][3] = 0;
    const T a = T(0.5) * qw, b = T(0.5) * qx, d = T(0.5) * qy, e = T(0.5) * qz;
    J.Fqw[0][0] = -b; J.Fqw[0][1] = -d; J.Fqw[0][2] = -e;
    J.Fqw[1][0] = a;  J.Fqw[1][1] = -e; J.Fqw[1][2] = d;
    J.Fqw[2][0] = e;  J.Fqw[2][1] = a;  J.Fqw[2][2] = -b;
    J.Fqw[3][0] = -d; J.Fqw[3][1] = b;  J.Fqw[3][2] = a;
    const T kx = -(c.J[2] - c.J[1]) * c.invJ[0], ky = -(c.J[0] - c.J[2]) * c.invJ[1],
            kz = -(c.J[1] - c.J[0]) * c.invJ[2];
    J.Fww[0][0] = 0;       J.Fww[0][1] = kx * wz; J.Fww[0][2] = kx * wy;
    J.Fww[1][0] = ky * wz; J.Fww[1][1] = 0;       J.Fww[1][2] = ky * wx;
    J.Fww[2][0] = kz * wy; J.Fww[2][1] = kz * wx; J.Fww[2][2] = 0;
    J.r3m[0] = T(2) * (qx * qz + qw * qy) * c.inv_mass;
    J.r3m[1] = T(2) * (qy * qz - qw * qx) * c.inv_mass;
    J.r3m[2] = (T(1) - T(2) * (qx * qx + qy * qy)) * c.inv_mass;
}

// one column (13 entries) of the forward-VDE right-hand side  F*s + g.
// CLS: 0 = q column (no omega rows), 1 = omega column, 2 = input column j (adds f_u[:,j]).
template <class T, int CLS>
NMPC_HD void vde_col(const Consts<T> &c, const Jac<T> &J, const T *s, T *k, int j)
{
    NMPC_UNROLL for (int i = 0; i < 3; i++) k[i] = s[3 + i];
    NMPC_UNROLL for (int i = 0; i < 3; i++) {
        T a = (CLS == 2) ? J.r3m[i] : T(0);
        NMPC_UNROLL for (int l = 0; l < 4; l++) a += J.Fvq[i][l] * s[6 + l];
        k[3 + i] = a;
    }
    NMPC_UNROLL for (int i = 0; i < 4; i++) {
        T a = 0;
        NMPC_UNROLL for (int l = 0; l < 4; l++) a += J.Fqq[i][l] * s[6 + l];
        if (CLS != 0) {
            NMPC_UNROLL for (int l = 0; l < 3; l++) a += J.Fqw[i][l] * s[10 + l];
        }
        k[6 + i] = a;
    }
    if (CLS != 0) {
        NMPC_UNROLL for (int i = 0; i < 3; i++) {
            T a = (CLS == 2) ? c.fuw[i][j] : T(0);
            NMPC_UNROLL for (int l = 0; l < 3; l++) a += J.Fww[i][l] * s[10 + l];
            k[10 + i] = a;
        }
    } else {
        NMPC_UNROLL for (int i = 0; i < 3; i++) k[10 + i] = 0;
    }
}

template <class T, int CLS>
NMPC_HD void midpoint_col(const Consts<T> &c, const Jac<T> &J1, const Jac<T> &J2, T *s, int j)
{
    T k1[NX], sm[NX], k2[NX];
    vde_col<T, CLS>(c, J1, s, k1, j);
    NMPC_UNROLL for (int i = 0; i < NX; i++) sm[i] = s[i] + T(0.5) * c.h * k1[i];
    vde_col<T, CLS>(c, J2, sm, k2, j);
    NMPC_UNROLL for (int i = 0; i < NX; i++) s[i] += c.h * k2[i];
}

// ERK (explicit midpoint x steps) with forward sensitivities for the 11 non-trivial columns.
// S[c][13]: c = 0..3 q, 4..6 omega, 7..10 u.   xn = phi(x,u).
template <class T>
NMPC_HD void erk_sens(const Consts<T> &c, const T *x, const T *u, T *xn, T S[11][NX])
{
    NMPC_UNROLL for (int cc = 0; cc < 11; cc++) {
        NMPC_UNROLL for (int i = 0; i < NX; i++) S[cc][i] = 0;
    }
    NMPC_UNROLL for (int cc = 0; cc < 7; cc++) S[cc][6 + cc] = 1;
    NMPC_UNROLL for (int i = 0; i < NX; i++) xn[i] = x[i];
    for (int st = 0; st < c.steps; st++) {
        Jac<T> J1, J2;
        T f1[NX], xm[NX], f2[NX];
        model_f(c, xn, u, f1);
        model_jac(c, xn, u, J1);
        NMPC_UNROLL for (int i = 0; i < NX; i++) xm[i] = xn[i] + T(0.5) * c.h * f1[i];
        model_f(c, xm, u, f2);
        model_jac(c, xm, u, J2);
        NMPC_UNROLL for (int i = 0; i < NX; i++) xn[i] += c.h * f2[i];
        NMPC_UNROLL for (int cc = 0; cc < 4; cc++) midpoint_col<T, 0>(c, J1, J2, S[cc], 0);
        NMPC_UNROLL for (int cc = 4; cc < 7; cc++) midpoint_col<T, 1>(c, J1, J2, S[cc], 0);
        NMPC_UNROLL for (int cc = 7; cc < 11; cc++) midpoint_col<T, 2>(c, J1, J2, S[cc], cc - 7);
    }
}

// ---------------------------------------------------------------------------------------------------
// Adjoint sensitivities (the CasADi-generated `expl_vde_adj` of the reference's model,
// controller.py:267-355).  model_adj is the continuous adjoint right-hand side
//     ( f_x(x,u)' lam , f_u(x,u)' lam )
// from the hand-derived sparse Jacobian blocks of model_jac (p rows: d pdot / d v = I; v rows: Fvq and the
// thrust direction r3m; q rows: Fqq, Fqw; omega rows: Fww and the rotor geometry fuw); erk_adjoint is its
// discrete counterpart: the reverse sweep through the controller's integrator (explicit midpoint x steps),
// i.e. [A B]' lam of one shooting interval WITHOUT forming A and B - what a reverse-mode (adjoint) RTI would
// use, and what the stationarity report below is built on.  In Gauss-Newton mode acados does not call
// expl_vde_adj on the solve path (it multiplies with the stored [B A]'), so neither do the solve kernels.
template <class T>
NMPC_HD void model_adj(const Consts<T> &c, const Jac<T> &J, const T *lam, T *ax /*13*/, T *au /*4*/)
{
    NMPC_UNROLL for (int i = 0; i < 3; i++) { ax[i] = 0; ax[3 + i] = lam[i]; }
    NMPC_UNROLL for (int l = 0; l < 4; l++) {
        T a = 0;
        NMPC_UNROLL for (int i = 0; i < 3; i++) a += J.Fvq[i][l] * lam[3 + i];
        NMPC_UNROLL for (int i = 0; i < 4; i++) a += J.Fqq[i][l] * lam[6 + i];
        ax[6 + l] = a;
    }
    NMPC_UNROLL for (int l = 0; l < 3; l++) {
        T a = 0;
        NMPC_UNROLL for (int i = 0; i < 4; i++) a += J.Fqw[i][l] * lam[6 + i];
        NMPC_UNROLL for (int i = 0; i < 3; i++) a += J.Fww[i][l] * lam[10 + i];
        ax[10 + l] = a;
    }
    const T tv = J.r3m[0] * lam[3] + J.r3m[1] * lam[4] + J.r3m[2] * lam[5];
    NMPC_UNROLL for (int j = 0; j < NU; j++)
        au[j] = tv + c.fuw[0][j] * lam[10] + c.fuw[1][j] * lam[11] + c.fuw[2][j] * lam[12];
}

// x+ = phi(x,u) (explicit midpoint x c.steps), then lam <- A' lam, gu <- B' lam by the reverse sweep.  MAXS = steps kept.
template <class T, int MAXS>
NMPC_HD void erk_adjoint(const Consts<T> &c, const T *x, const T *u, T *lam /*in: adjoint of x+, out: A' lam*/, T *gu /*B' lam*/,
                         T *xnext /*nullable*/)
{
    T xs[MAXS][NX], xm[MAXS][NX], xc[NX];
    NMPC_UNROLL for (int i = 0; i < NX; i++) xc[i] = x[i];
    const int ns = c.steps < MAXS ? c.steps : MAXS;
    for (int st = 0; st < ns; st++) {
        T f1[NX], f2[NX];
        model_f(c, xc, u, f1);
        NMPC_UNROLL for (int i = 0; i < NX; i++) { xs[st][i] = xc[i]; xm[st][i] = xc[i] + T(0.5) * c.h * f1[i]; }
        model_f(c, xm[st], u, f2);
        NMPC_UNROLL for (int i = 0; i < NX; i++) xc[i] += c.h * f2[i];
    }
    if (xnext) { NMPC_UNROLL for (int i = 0; i < NX; i++) xnext[i] = xc[i]; }
    NMPC_UNROLL for (int j = 0; j < NU; j++) gu[j] = 0;
    for (int st = ns - 1; st >= 0; st--) {
        // x+ = x + h f(xm,u), xm = x + h/2 f(x,u):  lam_m = h f_x(xm)' lam+,  lam = lam+ + lam_m + h/2 f_x(x)' lam_m
        Jac<T> J;
        T lm[NX], ax[NX], au[NU];
        model_jac(c, xm[st], u, J);
        model_adj(c, J, lam, ax, au);
        NMPC_UNROLL for (int i = 0; i < NX; i++) lm[i] = c.h * ax[i];
        NMPC_UNROLL for (int j = 0; j < NU; j++) gu[j] += c.h * au[j];
        model_jac(c, xs[st], u, J);
        model_adj(c, J, lm, ax, au);
        NMPC_UNROLL for (int i = 0; i < NX; i++) lam[i] += lm[i] + T(0.5) * c.h * ax[i];
        NMPC_UNROLL for (int j = 0; j < NU; j++) gu[j] += T(0.5) * c.h * au[j];
    }
}

constexpr int ADJ_MAX_STEPS = 4;


// ---------------------------------------------------------------------------------------
// Riccati stage helpers.  Ad[79] (column-packed), Bm[13][4].
template <class T>
NMPC_HD void load_ad(const T *AB, int Bp, int lane, T *Ad)
{
    NMPC_UNROLL for (int i = 0; i < AD_SIZE; i++) Ad[i] = NMPC_LD(AB, i);
}
template <class T>
NMPC_HD void load_b(const T *AB, int Bp, int lane, T Bm[NX][NU])
{
    NMPC_UNROLL for (int i = 0; i < NX; i++) {
        NMPC_UNROLL for (int j = 0; j < NU; j++) Bm[i][j] = NMPC_LD(AB, AD_SIZE + i * NU + j);
    }
}

// y = A' h   (structured)
template <class T>
NMPC_HD void at_mul(const Consts<T> &c, const T *Ad, const T *h, T *y)
{
    NMPC_UNROLL for (int i = 0; i < 3; i++) { y[i] = h[i]; y[3 + i] = c.dt * h[i] + h[3 + i]; }
    NMPC_UNROLL for (int cc = 0; cc < NZ; cc++) {
        T a = 0;
        NMPC_UNROLL for (int r = 0; r < ad_rows(cc); r++) a += Ad[ad_ofs(cc) + r] * h[r];
        y[6 + cc] = a;
    }
}

// y = A x + B u (+ y0)   (structured)
template <class T>
NMPC_HD void a_mul_add(const Consts<T> &c, const T *Ad, const T Bm[NX][NU], const T *x, const T *u, T *y)
{
    // y holds the offset (b or 0) on entry
    NMPC_UNROLL for (int i = 0; i < 3; i++) { y[i] += x[i] + c.dt * x[3 + i]; y[3 + i] += x[3 + i]; }
    NMPC_UNROLL for (int cc = 0; cc < NZ; cc++) {
        NMPC_UNROLL for (int r = 0; r < ad_rows(cc); r++) y[r] += Ad[ad_ofs(cc) + r] * x[6 + cc];
    }
    NMPC_UNROLL for (int i = 0; i < NX; i++) {
        NMPC_UNROLL for (int j = 0; j < NU; j++) y[i] += Bm[i][j] * u[j];
    }
}

// forward substitution with the stored factor: Lf[lidx(i,j)] for j<i, Lf[lidx(i,i)] = 1/L_ii
template <class T>
NMPC_HD void l_solve(const T *Lf, T *v)
{
    NMPC_UNROLL for (int i = 0; i < NU; i++) {
        T a = v[i];
        NMPC_UNROLL for (int j = 0; j < i; j++) a -= Lf[lidx(i, j)] * v[j];
        v[i] = a * Lf[lidx(i, i)];
    }
}
template <class T>
NMPC_HD void lt_solve(const T *Lf, T *v)
{
    NMPC_UNROLL for (int i = NU - 1; i >= 0; i--) {
        T a = v[i];
        NMPC_UNROLL for (int j = i + 1; j < NU; j++) a -= Lf[lidx(j, i)] * v[j];
        v[i] = a * Lf[lidx(i, i)];
    }
}

NMPC_HD double nmpc_rsqrt(double v) { return 1.0 / sqrt(v); }
NMPC_HD float nmpc_rsqrt(float v) { return 1.0f / sqrtf(v); }

// Backward Riccati stage with factorisation.
//   in : P (91, symmetric packed) = P_{k+1}, pv = p_{k+1}, D[4] = R + sigma, rhat[4]
//   out: P, pv of stage k (skipped when `first` -- stage 0 has delta x_0 = 0); L, M, m stored.
//   returns false if a pivot was not positive.
template <class T>
NMPC_HD bool ricc_factor_stage(const Consts<T> &c, T *P, T *pv, const T *ABk, const T *bk,
                               const T *qk, const T *D, const T *rhat, T *LMk, int Bp, int lane,
                               bool first, const T *ush = nullptr)
{
    // ush (nullable): inputs of the interior point's iterate - the stage is then solved for the input STEP w, u = ush + w:
    // x+ = A x + B w + (b + B ush); rhat is the caller's business (r + R ush + ...)
    bool ok = true;
    T Bm[NX][NU], PB[NX][NU], h[NX];
    load_b(ABk, Bp, lane, Bm);
    // PB = P B ; h = P b + p
    NMPC_UNROLL for (int i = 0; i < NX; i++) {
        NMPC_UNROLL for (int j = 0; j < NU; j++) {
            T a = 0;
            NMPC_UNROLL for (int l = 0; l < NX; l++) a += P[sidx(i, l)] * Bm[l][j];
            PB[i][j] = a;
        }
    }
    {
        T bb[NX];
        NMPC_UNROLL for (int i = 0; i < NX; i++) bb[i] = NMPC_LD(bk, i);
        if (ush) {
            NMPC_UNROLL for (int i = 0; i < NX; i++) {
                NMPC_UNROLL for (int j = 0; j < NU; j++) bb[i] += Bm[i][j] * ush[j];
            }
        }
        NMPC_UNROLL for (int i = 0; i < NX; i++) {
            T a = pv[i];
            NMPC_UNROLL for (int l = 0; l < NX; l++) a += P[sidx(i, l)] * bb[l];
            h[i] = a;
        }
    }
    // Huu = D + B'PB (lower), gu = rhat + B'h
    T Lf[10], mv[NU];
    NMPC_UNROLL for (int i = 0; i < NU; i++) {
        NMPC_UNROLL for (int j = 0; j <= i; j++) {
            T a = (i == j) ? D[i] : T(0);
            NMPC_UNROLL for (int l = 0; l < NX; l++) a += Bm[l][i] * PB[l][j];
            Lf[lidx(i, j)] = a;
        }
        T g = rhat[i];
        NMPC_UNROLL for (int l = 0; l < NX; l++) g += Bm[l][i] * h[l];
        mv[i] = g;
    }
    // Cholesky, diagonal stored inverted
    NMPC_UNROLL for (int j = 0; j < NU; j++) {
        T d = Lf[lidx(j, j)];
        NMPC_UNROLL for (int l = 0; l < j; l++) d -= Lf[lidx(j, l)] * Lf[lidx(j, l)];
        if (!(d > T(0))) { ok = false; d = T(1); }
        const T rd = nmpc_rsqrt(d);
        Lf[lidx(j, j)] = rd;
        NMPC_UNROLL for (int i = j + 1; i < NU; i++) {
            T a = Lf[lidx(i, j)];
            NMPC_UNROLL for (int l = 0; l < j; l++) a -= Lf[lidx(i, l)] * Lf[lidx(j, l)];
            Lf[lidx(i, j)] = a * rd;
        }
    }
    l_solve(Lf, mv);
    NMPC_UNROLL for (int i = 0; i < 10; i++) NMPC_ST(LMk, i, Lf[i]);
    NMPC_UNROLL for (int i = 0; i < NU; i++) NMPC_ST(LMk, 10 + NU * NX + i, mv[i]);

    T Ad[AD_SIZE];
    load_ad(ABk, Bp, lane, Ad);
    // M = L^{-1} (B' P A): column by column
    T M[NU][NX];
    NMPC_UNROLL for (int cc = 0; cc < NX; cc++) {
        T col[NU];
        NMPC_UNROLL for (int i = 0; i < NU; i++) {
            T a;
            if (cc < 3) a = PB[cc][i];
            else if (cc < 6) a = c.dt * PB[cc - 3][i] + PB[cc][i];
            else {
                a = 0;
                NMPC_UNROLL for (int r = 0; r < ad_rows(cc - 6); r++) a += PB[r][i] * Ad[ad_ofs(cc - 6) + r];
            }
            col[i] = a;
        }
        l_solve(Lf, col);
        NMPC_UNROLL for (int i = 0; i < NU; i++) { M[i][cc] = col[i]; NMPC_ST(LMk, 10 + i * NX + cc, col[i]); }
    }
    if (first) return ok;
    // p_k = q + A'h - M'm
    {
        T gx[NX];
        at_mul(c, Ad, h, gx);
        NMPC_UNROLL for (int i = 0; i < NX; i++) {
            T a = gx[i] + NMPC_LD(qk, i);
            NMPC_UNROLL for (int l = 0; l < NU; l++) a -= M[l][i] * mv[l];
            pv[i] = a;
        }
    }
    // T = P * Ad  (13 x 7)
    T Tm[NX][NZ];
    NMPC_UNROLL for (int cc = 0; cc < NZ; cc++) {
        NMPC_UNROLL for (int i = 0; i < NX; i++) {
            T a = 0;
            NMPC_UNROLL for (int r = 0; r < ad_rows(cc); r++) a += P[sidx(i, r)] * Ad[ad_ofs(cc) + r];
            Tm[i][cc] = a;
        }
    }
    // P_k = Q + A'PA - M'M, written in place in an order that only reads old entries
    // (v,v) then (p,v) then (p,p); the (.,z) blocks come from T.
    NMPC_UNROLL for (int i = 0; i < 3; i++) {
        NMPC_UNROLL for (int j = i; j < 3; j++) {
            T a = P[sidx(3 + i, 3 + j)] + c.dt * (P[sidx(i, 3 + j)] + P[sidx(j, 3 + i)]) + c.dt * c.dt * P[sidx(i, j)];
            if (i == j) a += c.Qd[3 + i];
            NMPC_UNROLL for (int l = 0; l < NU; l++) a -= M[l][3 + i] * M[l][3 + j];
            P[sidx(3 + i, 3 + j)] = a;
        }
    }
    {
        T pvb[3][3];
        NMPC_UNROLL for (int i = 0; i < 3; i++) {
            NMPC_UNROLL for (int j = 0; j < 3; j++) {
                T a = P[sidx(i, 3 + j)] + c.dt * P[sidx(i, j)];
                NMPC_UNROLL for (int l = 0; l < NU; l++) a -= M[l][i] * M[l][3 + j];
                pvb[i][j] = a;
            }
        }
        NMPC_UNROLL for (int i = 0; i < 3; i++) {
            NMPC_UNROLL for (int j = 0; j < 3; j++) P[sidx(i, 3 + j)] = pvb[i][j];
        }
    }
    NMPC_UNROLL for (int i = 0; i < 3; i++) {
        NMPC_UNROLL for (int j = i; j < 3; j++) {
            T a = P[sidx(i, j)];
            if (i == j) a += c.Qd[i];
            NMPC_UNROLL for (int l = 0; l < NU; l++) a -= M[l][i] * M[l][j];
            P[sidx(i, j)] = a;
        }
    }
    NMPC_UNROLL for (int cc = 0; cc < NZ; cc++) {
        NMPC_UNROLL for (int i = 0; i < 3; i++) {
            T a = Tm[i][cc], b2 = c.dt * Tm[i][cc] + Tm[3 + i][cc];
            NMPC_UNROLL for (int l = 0; l < NU; l++) { a -= M[l][i] * M[l][6 + cc]; b2 -= M[l][3 + i] * M[l][6 + cc]; }
            P[sidx(i, 6 + cc)] = a;
            P[sidx(3 + i, 6 + cc)] = b2;
        }
    }
    NMPC_UNROLL for (int ca = 0; ca < NZ; ca++) {
        NMPC_UNROLL for (int cb = ca; cb < NZ; cb++) {
            T a = (ca == cb) ? c.Qd[6 + ca] : T(0);
            NMPC_UNROLL for (int r = 0; r < ad_rows(ca); r++) a += Ad[ad_ofs(ca) + r] * Tm[r][cb];
            NMPC_UNROLL for (int l = 0; l < NU; l++) a -= M[l][6 + ca] * M[l][6 + cb];
            P[sidx(6 + ca, 6 + cb)] = a;
        }
    }
    return ok;
}

// Backward vector-only stage of the homogeneous solve (b = 0, q = 0): pv <- A'pv - M'm with
// m = L^{-1}(drhat + B'pv); m is stored over the slot of the affine m.
template <class T>
NMPC_HD void ricc_back_homog_stage(const Consts<T> &c, T *pv, const T *ABk, const T *drhat, T *LMk,
                                   int Bp, int lane, bool first)
{
    T Bm[NX][NU], Lf[10], mv[NU];
    load_b(ABk, Bp, lane, Bm);
    NMPC_UNROLL for (int i = 0; i < 10; i++) Lf[i] = NMPC_LD(LMk, i);
    NMPC_UNROLL for (int i = 0; i < NU; i++) {
        T g = drhat[i];
        NMPC_UNROLL for (int l = 0; l < NX; l++) g += Bm[l][i] * pv[l];
        mv[i] = g;
    }
    l_solve(Lf, mv);
    NMPC_UNROLL for (int i = 0; i < NU; i++) NMPC_ST(LMk, 10 + NU * NX + i, mv[i]);
    if (first) return;
    T Ad[AD_SIZE], gx[NX];
    load_ad(ABk, Bp, lane, Ad);
    at_mul(c, Ad, pv, gx);
    NMPC_UNROLL for (int i = 0; i < NX; i++) {
        T a = gx[i];
        NMPC_UNROLL for (int l = 0; l < NU; l++) a -= NMPC_LD(LMk, 10 + l * NX + i) * mv[l];
        pv[i] = a;
    }
}

// Forward stage: uh = -L^{-T}(M xh + m); xh <- A xh + B uh (+ b).  `zero_x`: xh is known to be 0.
template <class T>
NMPC_HD void ricc_forward_stage(const Consts<T> &c, T *xh, T *uh, const T *ABk, const T *bk,
                                const T *LMk, int Bp, int lane, bool with_b, bool zero_x, bool last, const T *ush = nullptr)
{
    // ush (nullable): uh is the input STEP from the iterate ush (see ricc_factor_stage); the state moves with ush + uh
    T Lf[10];
    NMPC_UNROLL for (int i = 0; i < 10; i++) Lf[i] = NMPC_LD(LMk, i);
    NMPC_UNROLL for (int i = 0; i < NU; i++) {
        T a = NMPC_LD(LMk, 10 + NU * NX + i);
        if (!zero_x) {
            NMPC_UNROLL for (int j = 0; j < NX; j++) a += NMPC_LD(LMk, 10 + i * NX + j) * xh[j];
        }
        uh[i] = -a;
    }
    lt_solve(Lf, uh);
    if (last) return;
    T Ad[AD_SIZE], Bm[NX][NU], y[NX];
    load_ad(ABk, Bp, lane, Ad);
    load_b(ABk, Bp, lane, Bm);
    NMPC_UNROLL for (int i = 0; i < NX; i++) y[i] = with_b ? NMPC_LD(bk, i) : T(0);
    T uu[NU];
    NMPC_UNROLL for (int i = 0; i < NU; i++) uu[i] = ush ? ush[i] + uh[i] : uh[i];
    a_mul_add(c, Ad, Bm, xh, uu, y);
    NMPC_UNROLL for (int i = 0; i < NX; i++) xh[i] = y[i];
}

}  // namespace nmpc
