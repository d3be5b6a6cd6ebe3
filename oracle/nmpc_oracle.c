/*
 * nmpc_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).  See nmpc_oracle.h.
 *
 * Readable, dense, unstructured FP64 restatement of what acados does for the OCP that
 * reference src/rotors_mpc_controller/controller.py:175-355 builds.  Deliberately written
 * with generic dense loops (no sparsity exploitation, AoS, one instance at a time) so
 * that it is an independent check of the structured SoA HIP kernels.
 *
 * [REF]      = follows /root/reference (file:line given).
 * [UPSTREAM] = restates published acados / HPIPM behaviour (un-vendored, unpinned).
 */
#include "nmpc_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_WARM_DELTA 1e-3   /* warm start of the interior point from a failed attempt: distance from the bounds / box width */
#define ORC_WARM_MU 1e-3      /* ... and the central-path value that floors its multipliers */
#define NX ORC_NX
#define NU ORC_NU
#define NY ORC_NY

/* ------------------------------------------------------------------------------------ */
/* Workspace memory.  A solve allocates ~10 N small arrays; done with malloc / free inside an OpenMP loop over instances the
 * allocator's locks serialise the threads (round 3: 2.6x on 256 cores).  orc_solve_batch_ex gives every thread ONE arena that
 * it resets per instance: an allocation is a pointer bump, a free is nothing, no lock is taken inside a solve.  Outside the
 * batch loop (single solves, the tests' direct calls) no arena is installed and these are malloc / free.  Not a change of the
 * arithmetic: only where the same arrays live.  (The sanitizer build keeps the heap, so ASan still sees every array.)          */
typedef struct { char *base; size_t cap, off, need; } orc_arena;
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define ORC_ASAN_CLANG 1
#endif
#endif
#if defined(__SANITIZE_ADDRESS__) || defined(ORC_ASAN_CLANG)    /* gcc | clang spelling of "built with -fsanitize=address" */
#define ORC_USE_ARENA 0
#else
#define ORC_USE_ARENA 1
#endif
#define ORC_ARENA_MAX ((size_t)64 << 20)
static _Thread_local orc_arena *t_arena = NULL;
static void *orc_malloc(size_t n)
{
    orc_arena *a = t_arena;
    if (a) {
        const size_t m = (n + 63) & ~(size_t)63;
        a->need += m;
        if (a->off + m <= a->cap) { void *p = a->base + a->off; a->off += m; return p; }
    }
    return malloc(n ? n : 1);
}
static void *orc_calloc(size_t k, size_t sz)
{
    void *p = orc_malloc(k * sz);
    if (p) memset(p, 0, k * sz);
    return p;
}
static void orc_free(void *p)
{
    const orc_arena *a = t_arena;
    if (a && (char *)p >= a->base && (char *)p < a->base + a->cap) return;   /* arena memory: released by the reset */
    free(p);
}

/* ------------------------------------------------------------------------------------ */
/* defaults: reference config/params.yaml:1-33 and controller.py:98-110                  */
void orc_default_config(orc_config *c)
{
    static const double W[NY] = {10, 10, 8, 1, 1, 0.2, 3.2, 3.2, 3.2, 3.2, 1.4, 1.4, 0.4,
                                 1.75, 1.75, 1.75, 1.75};
    static const double We[NX] = {5, 5, 3, 2, 2, 2, 12, 12, 12, 18.5, 2, 2, 1.8};
    const double L = 0.17, kf = 8.54858e-6, km = 0.016;
    const double spin[NU] = {-1.0, 1.0, -1.0, 1.0};
    const double rx[NU] = {L, 0.0, -L, 0.0}, ry[NU] = {0.0, L, 0.0, -L};
    memset(c, 0, sizeof(*c));
    c->N = 20;
    c->dt = 0.05;
    memcpy(c->W, W, sizeof(W));
    memcpy(c->We, We, sizeof(We));
    for (int i = 0; i < NU; i++) {
        c->lbu[i] = kf * 50.0 * 50.0;   /* controller.py:105 */
        c->ubu[i] = kf * 838.0 * 838.0; /* controller.py:106 */
        c->rotor_x[i] = rx[i];
        c->rotor_y[i] = ry[i];
        c->rotor_z[i] = spin[i] * km;   /* controller.py:103 */
    }
    c->lm = 7.0e-3;
    c->lm_scaled_by_dt = 1;
    c->cost_scaled_by_dt = 1;
    c->mass = 0.68;
    c->gravity = 9.81;
    c->J[0] = 0.007; c->J[1] = 0.007; c->J[2] = 0.012;
    c->sim_num_stages = 2;
    c->sim_num_steps = 2;
    c->qp_iter_max = 600;
    c->qp_cond_N = 0;
    c->qp_tol_comp = 1e-11;
    c->qp_tol_stat = 1e-11;
    c->qp_mu0 = 0.1;
    c->qp_tau = 0.995;
    c->qp_thr0 = 0.1;
    c->qp_thr0_rel = 0.25;
    c->qp_gamma = 0.0;   /* optional safeguard; the HIP kernels do not implement it */
    c->qp_polish = 0;
    c->qp_polish_mu = 1.0;      /* >= mu0: the first attempt is a pure active-set solve from 'all free' */
    c->qp_polish_passes = 0;   /* 0 = the GPU library's default attempt policy for the horizon (orc_polish_policy below) */
    c->qp_polish_budget = 0;
    c->qp_growth_max = 1e6;    /* ~1e-15 * growth of relative accuracy is lost: 1e-9 is still held */
    c->qp_acc_comp = 1e-8;     /* [UPSTREAM] HPIPM's default res_m_max */
    c->qp_acc_stat = 1e-8;
    c->qp_tol_step = 1e-3;
    c->qp_maxiter_status = 0;
    c->qp_warm_start = 1;
    c->qp_exit_mode = 0;
    c->qp_bound_res = 0;
}

/* ------------------------------------------------------------------------------------ */
/* model, [REF] controller.py:267-355                                                    */
void orc_model_f(const orc_config *c, const double *x, const double *u, double *f)
{
    const double vx = x[3], vy = x[4], vz = x[5];
    const double qw = x[6], qx = x[7], qy = x[8], qz = x[9];
    const double wx = x[10], wy = x[11], wz = x[12];
    /* third column of the rotation matrix, controller.py:296,299,302 */
    const double r13 = 2.0 * (qx * qz + qw * qy);
    const double r23 = 2.0 * (qy * qz - qw * qx);
    const double r33 = 1.0 - 2.0 * (qx * qx + qy * qy);
    const double T = (u[0] + u[1] + u[2] + u[3]) / c->mass; /* controller.py:310-313 */
    f[0] = vx; f[1] = vy; f[2] = vz;
    f[3] = r13 * T;
    f[4] = r23 * T;
    f[5] = r33 * T - c->gravity;                             /* controller.py:314 */
    f[6] = 0.5 * (-qx * wx - qy * wy - qz * wz);             /* controller.py:316 */
    f[7] = 0.5 * (qw * wx + qy * wz - qz * wy);              /* controller.py:317 */
    f[8] = 0.5 * (qw * wy + qz * wx - qx * wz);              /* controller.py:318 */
    f[9] = 0.5 * (qw * wz + qx * wy - qy * wx);              /* controller.py:319 */
    double tx = 0, ty = 0, tz = 0;                           /* controller.py:326-328 */
    for (int i = 0; i < NU; i++) {
        tx += u[i] * c->rotor_y[i];
        ty += u[i] * (-c->rotor_x[i]);
        tz += u[i] * c->rotor_z[i];
    }
    const double Jx = c->J[0], Jy = c->J[1], Jz = c->J[2];
    const double cx = wy * (Jz * wz) - wz * (Jy * wy);       /* controller.py:335 */
    const double cy = wz * (Jx * wx) - wx * (Jz * wz);       /* controller.py:336 */
    const double cz = wx * (Jy * wy) - wy * (Jx * wx);       /* controller.py:337 */
    f[10] = (tx - cx) / Jx;                                  /* controller.py:339-341 */
    f[11] = (ty - cy) / Jy;
    f[12] = (tz - cz) / Jz;
}

/* hand-derived Jacobians of orc_model_f (checked against sympy in tests, K4) */
void orc_model_jac(const orc_config *c, const double *x, const double *u, double *fx, double *fu)
{
    const double qw = x[6], qx = x[7], qy = x[8], qz = x[9];
    const double wx = x[10], wy = x[11], wz = x[12];
    const double T = (u[0] + u[1] + u[2] + u[3]) / c->mass;
    const double Jx = c->J[0], Jy = c->J[1], Jz = c->J[2];
    memset(fx, 0, sizeof(double) * NX * NX);
    memset(fu, 0, sizeof(double) * NX * NU);
#define FX(r, cc) fx[(r) * NX + (cc)]
#define FU(r, cc) fu[(r) * NU + (cc)]
    FX(0, 3) = 1.0; FX(1, 4) = 1.0; FX(2, 5) = 1.0;
    /* d vdot / d q */
    FX(3, 6) = 2.0 * qy * T;  FX(3, 7) = 2.0 * qz * T;  FX(3, 8) = 2.0 * qw * T;  FX(3, 9) = 2.0 * qx * T;
    FX(4, 6) = -2.0 * qx * T; FX(4, 7) = -2.0 * qw * T; FX(4, 8) = 2.0 * qz * T;  FX(4, 9) = 2.0 * qy * T;
    FX(5, 7) = -4.0 * qx * T; FX(5, 8) = -4.0 * qy * T;
    /* d qdot / d q */
    FX(6, 7) = -0.5 * wx; FX(6, 8) = -0.5 * wy; FX(6, 9) = -0.5 * wz;
    FX(7, 6) = 0.5 * wx;  FX(7, 8) = 0.5 * wz;  FX(7, 9) = -0.5 * wy;
    FX(8, 6) = 0.5 * wy;  FX(8, 7) = -0.5 * wz; FX(8, 9) = 0.5 * wx;
    FX(9, 6) = 0.5 * wz;  FX(9, 7) = 0.5 * wy;  FX(9, 8) = -0.5 * wx;
    /* d qdot / d omega */
    FX(6, 10) = -0.5 * qx; FX(6, 11) = -0.5 * qy; FX(6, 12) = -0.5 * qz;
    FX(7, 10) = 0.5 * qw;  FX(7, 11) = -0.5 * qz; FX(7, 12) = 0.5 * qy;
    FX(8, 10) = 0.5 * qz;  FX(8, 11) = 0.5 * qw;  FX(8, 12) = -0.5 * qx;
    FX(9, 10) = -0.5 * qy; FX(9, 11) = 0.5 * qx;  FX(9, 12) = 0.5 * qw;
    /* d omegadot / d omega : omegadot_x = (tx - (Jz-Jy) wy wz)/Jx etc. */
    FX(10, 11) = -(Jz - Jy) * wz / Jx; FX(10, 12) = -(Jz - Jy) * wy / Jx;
    FX(11, 10) = -(Jx - Jz) * wz / Jy; FX(11, 12) = -(Jx - Jz) * wx / Jy;
    FX(12, 10) = -(Jy - Jx) * wy / Jz; FX(12, 11) = -(Jy - Jx) * wx / Jz;
    const double r13 = 2.0 * (qx * qz + qw * qy);
    const double r23 = 2.0 * (qy * qz - qw * qx);
    const double r33 = 1.0 - 2.0 * (qx * qx + qy * qy);
    for (int i = 0; i < NU; i++) {
        FU(3, i) = r13 / c->mass;
        FU(4, i) = r23 / c->mass;
        FU(5, i) = r33 / c->mass;
        FU(10, i) = c->rotor_y[i] / Jx;
        FU(11, i) = -c->rotor_x[i] / Jy;
        FU(12, i) = c->rotor_z[i] / Jz;
    }
#undef FX
#undef FU
}

/* [UPSTREAM] counterpart of CasADi-generated <model>_expl_vde_forw:
 * (x, Sx, Su, u) -> (f, fx*Sx, fx*Su + fu)                                              */
void orc_vde_forw(const orc_config *c, const double *x, const double *Sx, const double *Su,
                  const double *u, double *xdot, double *Sxdot, double *Sudot)
{
    double fx[NX * NX], fu[NX * NU];
    orc_model_f(c, x, u, xdot);
    orc_model_jac(c, x, u, fx, fu);
    for (int i = 0; i < NX; i++) {
        for (int j = 0; j < NX; j++) {
            double s = 0.0;
            for (int k = 0; k < NX; k++) s += fx[i * NX + k] * Sx[k * NX + j];
            Sxdot[i * NX + j] = s;
        }
        for (int j = 0; j < NU; j++) {
            double s = fu[i * NU + j];
            for (int k = 0; k < NX; k++) s += fx[i * NX + k] * Su[k * NU + j];
            Sudot[i * NU + j] = s;
        }
    }
}

/* [UPSTREAM] counterpart of <model>_expl_vde_adj: jtimes(f, [x;u], lam, transpose) */
void orc_vde_adj(const orc_config *c, const double *x, const double *lam, const double *u,
                 double *adj)
{
    double fx[NX * NX], fu[NX * NU];
    orc_model_jac(c, x, u, fx, fu);
    for (int j = 0; j < NX; j++) {
        double s = 0.0;
        for (int i = 0; i < NX; i++) s += fx[i * NX + j] * lam[i];
        adj[j] = s;
    }
    for (int j = 0; j < NU; j++) {
        double s = 0.0;
        for (int i = 0; i < NX; i++) s += fu[i * NU + j] * lam[i];
        adj[NX + j] = s;
    }
}

/* ------------------------------------------------------------------------------------ */
/* [UPSTREAM] acados sim_erk: explicit RK on the forward VDE.  Tableaus as acados selects
 * them from sim_method_num_stages: 1 Euler, 2 explicit midpoint, 3 Kutta-3, 4 RK4.       */
static void erk_tableau(int ns, double *Arr /*ns*ns*/, double *brr, double *crr)
{
    memset(Arr, 0, sizeof(double) * 16);
    memset(brr, 0, sizeof(double) * 4);
    memset(crr, 0, sizeof(double) * 4);
    switch (ns) {
    case 1: brr[0] = 1.0; break;
    case 2: Arr[1 * ns + 0] = 0.5; brr[1] = 1.0; crr[1] = 0.5; break;
    case 3:
        Arr[1 * ns + 0] = 0.5; Arr[2 * ns + 0] = -1.0; Arr[2 * ns + 1] = 2.0;
        brr[0] = 1.0 / 6.0; brr[1] = 2.0 / 3.0; brr[2] = 1.0 / 6.0;
        crr[1] = 0.5; crr[2] = 1.0; break;
    default: /* 4 */
        Arr[1 * 4 + 0] = 0.5; Arr[2 * 4 + 1] = 0.5; Arr[3 * 4 + 2] = 1.0;
        brr[0] = 1.0 / 6.0; brr[1] = 1.0 / 3.0; brr[2] = 1.0 / 3.0; brr[3] = 1.0 / 6.0;
        crr[1] = 0.5; crr[2] = 0.5; crr[3] = 1.0; break;
    }
}

void orc_integrate(const orc_config *c, const double *x, const double *u,
                   double *xn, double *A, double *B)
{
    int ns = c->sim_num_stages;
    if (ns < 1 || ns > 4) ns = 4;
    const int steps = c->sim_num_steps > 0 ? c->sim_num_steps : 1;
    const double h = c->dt / steps;
    double Arr[16], brr[4], crr[4];
    erk_tableau(ns, Arr, brr, crr);
    enum { NS = NX + NX * NX + NX * NU };
    double z[NS], zs[NS], K[4][NS];
    memcpy(z, x, sizeof(double) * NX);
    memset(z + NX, 0, sizeof(double) * (NX * NX + NX * NU));
    for (int i = 0; i < NX; i++) z[NX + i * NX + i] = 1.0; /* Sx(0) = I, Su(0) = 0 */
    for (int st = 0; st < steps; st++) {
        for (int s = 0; s < ns; s++) {
            memcpy(zs, z, sizeof(z));
            for (int j = 0; j < s; j++) {
                const double a = h * Arr[s * ns + j];
                if (a != 0.0)
                    for (int i = 0; i < NS; i++) zs[i] += a * K[j][i];
            }
            orc_vde_forw(c, zs, zs + NX, zs + NX + NX * NX, u,
                         K[s], K[s] + NX, K[s] + NX + NX * NX);
        }
        for (int s = 0; s < ns; s++) {
            const double w = h * brr[s];
            if (w != 0.0)
                for (int i = 0; i < NS; i++) z[i] += w * K[s][i];
        }
    }
    memcpy(xn, z, sizeof(double) * NX);
    memcpy(A, z + NX, sizeof(double) * NX * NX);
    memcpy(B, z + NX + NX * NX, sizeof(double) * NX * NU);
}

/* ------------------------------------------------------------------------------------ */
/* [UPSTREAM] SQP_RTI preparation: dynamics linearisation + LINEAR_LS Gauss-Newton cost
 * (U3, U4), Levenberg-Marquardt (U5), PROJECT_REDUC_HESS as a check (U6), input bounds
 * shifted to the linearisation point.  Vx=[I;0], Vu=[0;I] (controller.py:226-235) make
 * y = [x;u] and every Hessian block diagonal.                                           */
void orc_linearize(const orc_config *c, const double *xtraj, const double *utraj,
                   const double *yref, const double *yref_e,
                   double *A, double *B, double *b, double *q, double *r,
                   double *lo, double *hi, double *Qd, double *Rd, int *hess_projected)
{
    const int N = c->N;
    int projected = 0;
    for (int k = 0; k < N; k++) {
        const double *xk = xtraj + k * NX, *uk = utraj + k * NU;
        double xn[NX];
        orc_integrate(c, xk, uk, xn, A + k * NX * NX, B + k * NX * NU);
        for (int i = 0; i < NX; i++) b[k * NX + i] = xn[i] - xtraj[(k + 1) * NX + i];
        const double sc = c->cost_scaled_by_dt ? c->dt : 1.0;
        const double lmk = c->lm * (c->lm_scaled_by_dt ? c->dt : 1.0);
        for (int i = 0; i < NX; i++) {
            q[k * NX + i] = sc * c->W[i] * (xk[i] - yref[k * NY + i]);
            Qd[k * NX + i] = sc * c->W[i] + lmk;
            if (!(Qd[k * NX + i] > 0.0)) projected = 1;
        }
        for (int i = 0; i < NU; i++) {
            r[k * NU + i] = sc * c->W[NX + i] * (uk[i] - yref[k * NY + NX + i]);
            Rd[k * NU + i] = sc * c->W[NX + i] + lmk;
            if (!(Rd[k * NU + i] > 0.0)) projected = 1;
            lo[k * NU + i] = c->lbu[i] - uk[i];
            hi[k * NU + i] = c->ubu[i] - uk[i];
        }
    }
    for (int i = 0; i < NX; i++) {
        q[N * NX + i] = c->We[i] * (xtraj[N * NX + i] - yref_e[i]);
        Qd[N * NX + i] = c->We[i] + c->lm;
        if (!(Qd[N * NX + i] > 0.0)) projected = 1;
    }
    if (hess_projected) *hess_projected = projected;
}

/* ------------------------------------------------------------------------------------ */
/* generic OCP-QP with box bounds on the inputs only (dense stage Hessians, variable nu)  */
typedef struct {
    int N;
    int *nu;
    double **A, **B, **b;   /* A nx*nx, B nx*nu (row-major) */
    double **Q, **R, **S;   /* Q nx*nx, R nu*nu, S nu*nx */
    double **q, **r, **lo, **hi;
    double *QN, *qN;        /* terminal nx*nx, nx */
} ocpqp;

static double *dalloc(size_t n) { return (double *)orc_calloc(n ? n : 1, sizeof(double)); }

static void ocpqp_alloc(ocpqp *p, int N, const int *nu)
{
    p->N = N;
    p->nu = (int *)orc_malloc(sizeof(int) * (size_t)N);
    p->A = (double **)orc_malloc(sizeof(double *) * N); p->B = (double **)orc_malloc(sizeof(double *) * N);
    p->b = (double **)orc_malloc(sizeof(double *) * N); p->Q = (double **)orc_malloc(sizeof(double *) * N);
    p->R = (double **)orc_malloc(sizeof(double *) * N); p->S = (double **)orc_malloc(sizeof(double *) * N);
    p->q = (double **)orc_malloc(sizeof(double *) * N); p->r = (double **)orc_malloc(sizeof(double *) * N);
    p->lo = (double **)orc_malloc(sizeof(double *) * N); p->hi = (double **)orc_malloc(sizeof(double *) * N);
    for (int k = 0; k < N; k++) {
        const int m = nu[k];
        p->nu[k] = m;
        p->A[k] = dalloc(NX * NX); p->B[k] = dalloc((size_t)NX * m); p->b[k] = dalloc(NX);
        p->Q[k] = dalloc(NX * NX); p->R[k] = dalloc((size_t)m * m); p->S[k] = dalloc((size_t)m * NX);
        p->q[k] = dalloc(NX); p->r[k] = dalloc(m); p->lo[k] = dalloc(m); p->hi[k] = dalloc(m);
    }
    p->QN = dalloc(NX * NX);
    p->qN = dalloc(NX);
}

static void ocpqp_free(ocpqp *p)
{
    for (int k = 0; k < p->N; k++) {
        orc_free(p->A[k]); orc_free(p->B[k]); orc_free(p->b[k]); orc_free(p->Q[k]); orc_free(p->R[k]); orc_free(p->S[k]);
        orc_free(p->q[k]); orc_free(p->r[k]); orc_free(p->lo[k]); orc_free(p->hi[k]);
    }
    orc_free(p->A); orc_free(p->B); orc_free(p->b); orc_free(p->Q); orc_free(p->R); orc_free(p->S);
    orc_free(p->q); orc_free(p->r); orc_free(p->lo); orc_free(p->hi); orc_free(p->nu); orc_free(p->QN); orc_free(p->qN);
}

/* in-place lower Cholesky of an m*m row-major SPD matrix.
 * A pivot is valid while 0 < d <= ORC_PIVOT_MAX; returns 0 ok | 1 the first invalid pivot is not positive (a failed factorisation) |
 * 2 it is NaN or beyond ORC_PIVOT_MAX - not-a-number data: a magnitude that can no longer be squared in double precision is one
 * (fuzz seed 11856, a warm start about a trajectory at |x| 5e10: pivot 8.9e269, then an exact 0 here and an inf - inf in the kernels'
 * L D L' - which of the two classes came out depended on whose arithmetic overflowed first; the kernels carry the same test and the
 * same constant, csrc/nmpc_team.hpp PIVOT_MAX) */
#define ORC_PIVOT_MAX 1e100
static int chol_lower(double *M, int m)
{
    for (int j = 0; j < m; j++) {
        double d = M[j * m + j];
        for (int k = 0; k < j; k++) d -= M[j * m + k] * M[j * m + k];
        if (!(fabs(d) <= ORC_PIVOT_MAX)) return 2;
        if (!(d > 0.0)) return 1;
        d = sqrt(d);
        M[j * m + j] = d;
        for (int i = j + 1; i < m; i++) {
            double s = M[i * m + j];
            for (int k = 0; k < j; k++) s -= M[i * m + k] * M[j * m + k];
            M[i * m + j] = s / d;
        }
        for (int i = 0; i < j; i++) M[i * m + j] = 0.0;
    }
    return 0;
}

/* y := L^{-1} y (forward substitution), ncol right-hand sides stored row-major m*ncol */
static void trsm_lower(const double *L, int m, double *Y, int ncol)
{
    for (int i = 0; i < m; i++) {
        for (int cidx = 0; cidx < ncol; cidx++) {
            double s = Y[i * ncol + cidx];
            for (int k = 0; k < i; k++) s -= L[i * m + k] * Y[k * ncol + cidx];
            Y[i * ncol + cidx] = s / L[i * m + i];
        }
    }
}

/* y := L^{-T} y for a single vector */
static void trsv_lower_t(const double *L, int m, double *y)
{
    for (int i = m - 1; i >= 0; i--) {
        double s = y[i];
        for (int k = i + 1; k < m; k++) s -= L[k * m + i] * y[k];
        y[i] = s / L[i * m + i];
    }
}

typedef struct {
    double **L, **M, **m; /* per stage: chol(Huu) nu*nu, L^{-1}Hux nu*nx, L^{-1}gu nu */
    double **P, **p;      /* optional (NULL or N+1 entries): the cost-to-go (P_k nx*nx, p_k nx) a factor pass leaves, k = 0..N */
} ricc_fact;

/* Backward Riccati sweep.  factor != 0: build L,M from D (= R + diag(sig)) and the data;
 * always: vector recursion with gradient rhat, offsets bb (NULL = 0), state gradient qq
 * (NULL = 0), terminal qN (NULL = 0).  Returns 0 ok, 1 factorisation failure (a pivot that is not positive), 2 the first pivot to
 * fail is a NaN: NaN data (the kernels make the same distinction: nmpc_stage.hpp pivot()).  */
static int riccati_backward(const ocpqp *p, double **sig, double **rhat, int homogeneous,
                            int factor, ricc_fact *f, double *gmax)
{
    /* gmax (factor pass, may be NULL): max over stages and entries of |B_k' P_{k+1} B_k| - the growth certificate */
    const int N = p->N;
    double P[NX * NX], pv[NX], PA[NX * NX], Hxx[NX * NX], h[NX], gx[NX];
    if (gmax) *gmax = 0.0;
    if (factor) memcpy(P, p->QN, sizeof(P));
    for (int i = 0; i < NX; i++) pv[i] = homogeneous ? 0.0 : p->qN[i];
    if (factor && f->P) { memcpy(f->P[N], P, sizeof(P)); memcpy(f->p[N], pv, sizeof(pv)); }
    /* In the factor pass P holds P_{k+1}.  In a vector-only pass (factor == 0) the
     * homogeneous recursion needs no P at all (b = 0).                                   */
    for (int k = N - 1; k >= 0; k--) {
        const int m = p->nu[k];
        const double *A = p->A[k], *B = p->B[k];
        double *L = f->L[k], *M = f->M[k], *mv = f->m[k];
        double *PB = dalloc((size_t)NX * m);
        double *gu = dalloc(m);
        /* h = P b + p */
        for (int i = 0; i < NX; i++) {
            double s = pv[i];
            if (!homogeneous)
                for (int j = 0; j < NX; j++) s += P[i * NX + j] * p->b[k][j];
            h[i] = s;
        }
        if (factor) {
            for (int i = 0; i < NX; i++) {
                for (int j = 0; j < NX; j++) {
                    double s = 0.0;
                    for (int l = 0; l < NX; l++) s += P[i * NX + l] * A[l * NX + j];
                    PA[i * NX + j] = s;
                }
                for (int j = 0; j < m; j++) {
                    double s = 0.0;
                    for (int l = 0; l < NX; l++) s += P[i * NX + l] * B[l * m + j];
                    PB[i * m + j] = s;
                }
            }
            /* Huu = R + diag(sig) + B' P B */
            for (int i = 0; i < m; i++)
                for (int j = 0; j < m; j++) {
                    double s = 0.0;
                    for (int l = 0; l < NX; l++) s += B[l * m + i] * PB[l * m + j];
                    if (gmax && fabs(s) > *gmax) *gmax = fabs(s);
                    if (gmax && !(s == s)) *gmax = s;              /* NaN stays NaN */
                    L[i * m + j] = p->R[k][i * m + j] + (i == j ? sig[k][i] : 0.0) + s;
                }
            /* Hux = S + B' P A */
            for (int i = 0; i < m; i++)
                for (int j = 0; j < NX; j++) {
                    double s = p->S[k][i * NX + j];
                    for (int l = 0; l < NX; l++) s += B[l * m + i] * PA[l * NX + j];
                    M[i * NX + j] = s;
                }
            /* Hxx = Q + A' P A */
            for (int i = 0; i < NX; i++)
                for (int j = 0; j < NX; j++) {
                    double s = p->Q[k][i * NX + j];
                    for (int l = 0; l < NX; l++) s += A[l * NX + i] * PA[l * NX + j];
                    Hxx[i * NX + j] = s;
                }
            { const int cf = chol_lower(L, m); if (cf) { orc_free(PB); orc_free(gu); return cf; } }
            trsm_lower(L, m, M, NX);
            /* P_k = Hxx - M'M, symmetrised */
            for (int i = 0; i < NX; i++)
                for (int j = 0; j < NX; j++) {
                    double s = Hxx[i * NX + j];
                    for (int l = 0; l < m; l++) s -= M[l * NX + i] * M[l * NX + j];
                    PA[i * NX + j] = s;
                }
            for (int i = 0; i < NX; i++)
                for (int j = 0; j < NX; j++) P[i * NX + j] = 0.5 * (PA[i * NX + j] + PA[j * NX + i]);
        }
        /* gu = rhat + B'h ; gx = q + A'h */
        for (int i = 0; i < m; i++) {
            double s = rhat[k][i];
            for (int l = 0; l < NX; l++) s += B[l * m + i] * h[l];
            gu[i] = s;
        }
        for (int i = 0; i < NX; i++) {
            double s = homogeneous ? 0.0 : p->q[k][i];
            for (int l = 0; l < NX; l++) s += A[l * NX + i] * h[l];
            gx[i] = s;
        }
        trsm_lower(L, m, gu, 1);
        memcpy(mv, gu, sizeof(double) * m);
        for (int i = 0; i < NX; i++) {
            double s = gx[i];
            for (int l = 0; l < m; l++) s -= M[l * NX + i] * mv[l];
            pv[i] = s;
        }
        if (factor && f->P) { memcpy(f->P[k], P, sizeof(P)); memcpy(f->p[k], pv, sizeof(pv)); }
        orc_free(PB);
        orc_free(gu);
    }
    return 0;
}

/* forward sweep: xh_0 = dx0 (or 0), uh_k = -L^{-T}(M xh_k + m), xh_{k+1} = A xh + B uh + b */
static void riccati_forward(const ocpqp *p, const ricc_fact *f, const double *dx0,
                            int homogeneous, double **uh, double *xh /*(N+1)*NX*/)
{
    const int N = p->N;
    for (int i = 0; i < NX; i++) xh[i] = (homogeneous || !dx0) ? 0.0 : dx0[i];
    for (int k = 0; k < N; k++) {
        const int m = p->nu[k];
        const double *xk = xh + k * NX;
        double *xn = xh + (k + 1) * NX;
        for (int i = 0; i < m; i++) {
            double s = f->m[k][i];
            for (int j = 0; j < NX; j++) s += f->M[k][i * NX + j] * xk[j];
            uh[k][i] = -s;
        }
        trsv_lower_t(f->L[k], m, uh[k]);
        for (int i = 0; i < NX; i++) {
            double s = homogeneous ? 0.0 : p->b[k][i];
            for (int j = 0; j < NX; j++) s += p->A[k][i * NX + j] * xk[j];
            for (int j = 0; j < m; j++) s += p->B[k][i * m + j] * uh[k][j];
            xn[i] = s;
        }
    }
}

/* Active-set polish of an interior-point iterate (not in HPIPM; an exactness/latency device of this
 * build, mirrored by the team kernel).  Classify every bound pair from the current iterate (a bound
 * is taken as active when its multiplier exceeds its slack), pin those inputs at their bounds, solve
 * the remaining equality-constrained LQ problem exactly with one Riccati factorisation, and ACCEPT
 * the result only if it satisfies the KKT conditions of the original QP: free inputs inside their
 * box and multipliers of the pinned inputs of the right sign (costates by the adjoint recursion).
 * An accepted point is THE solution of the strictly convex QP; a rejected one costs one sweep and the
 * IPM simply continues.  Returns 1 if accepted (u, x overwritten; ll, lu set to the multipliers).    */
static int ocpqp_polish(const ocpqp *p, const double *dx0, const ricc_fact *f, double **u, double **ll,
                        double **lu, double **tlo, double **tup, double *x, int max_pass, int *passes, double growth_max,
                        double *gbase, double *growth, int *untrusted, int *warm)
{
    /* tlo, tup: the slacks the interior point carries (ocpqp_ipm); the active-set guess compares a multiplier with ITS slack */
    /* warm (nullable): out, 1 when the attempt ran out of passes and (u, ll, lu) were replaced by the warm start built from its
     * last pass (the pass must have had pins: a pass from "all free" that fails leaves nothing to build on)                    */
    int exhausted = 0;
    /* growth certificate (orc_config.qp_growth_max): *gbase = g of the first factorisation of the solve (0 = none yet: this
     * call's first pass sets it); a pass whose g exceeds growth_max * *gbase ends the attempt unaccepted, *untrusted = 1 */
    const int N = p->N;
    ocpqp m = *p;
    m.B = (double **)orc_malloc(sizeof(double *) * N); m.b = (double **)orc_malloc(sizeof(double *) * N);
    m.R = (double **)orc_malloc(sizeof(double *) * N); m.S = (double **)orc_malloc(sizeof(double *) * N);
    m.q = (double **)orc_malloc(sizeof(double *) * N); m.r = (double **)orc_malloc(sizeof(double *) * N);
    double **zero = (double **)orc_malloc(sizeof(double *) * N), **uh = (double **)orc_malloc(sizeof(double *) * N);
    int **pin = (int **)orc_malloc(sizeof(int *) * N), **newpin = (int **)orc_malloc(sizeof(int *) * N);
    double **gsave = (double **)orc_malloc(sizeof(double *) * N);
    double *xh = dalloc((size_t)(N + 1) * NX);
    /* the cost-to-go of every pass is kept: the multiplier of a pinned input is formed with the costate P_{k+1} x_{k+1} + p_{k+1}.
     * (The adjoint recursion pi_k = Q x + q + A'pi_{k+1} gives the same number in exact arithmetic but amplifies rounding by
     * rho(A) per stage: on a violently unstable plant - rho = 2, N = 120 - it accepted active sets whose multipliers, solved in
     * 60 digits, had the wrong sign by 8e-2, where the tile kernels' P-based check went on to the right set.)                  */
    ricc_fact fp = *f;
    fp.P = (double **)orc_malloc(sizeof(double *) * (size_t)(N + 1)); fp.p = (double **)orc_malloc(sizeof(double *) * (size_t)(N + 1));
    for (int k = 0; k <= N; k++) { fp.P[k] = dalloc(NX * NX); fp.p[k] = dalloc(NX); }
    for (int k = 0; k < N; k++) {
        const int nu = p->nu[k];
        m.B[k] = dalloc((size_t)NX * nu); m.b[k] = dalloc(NX); m.R[k] = dalloc((size_t)nu * nu);
        m.S[k] = dalloc((size_t)nu * NX); m.q[k] = dalloc(NX); m.r[k] = dalloc(nu);
        zero[k] = dalloc(nu); uh[k] = dalloc(nu); pin[k] = (int *)orc_calloc((size_t)nu, sizeof(int));
        newpin[k] = (int *)orc_calloc((size_t)nu, sizeof(int)); gsave[k] = dalloc(nu);
        memcpy(m.B[k], p->B[k], sizeof(double) * NX * nu); memcpy(m.b[k], p->b[k], sizeof(double) * NX);
        memcpy(m.R[k], p->R[k], sizeof(double) * nu * nu); memcpy(m.S[k], p->S[k], sizeof(double) * nu * NX);
        memcpy(m.q[k], p->q[k], sizeof(double) * NX); memcpy(m.r[k], p->r[k], sizeof(double) * nu);
        for (int i = 0; i < nu; i++) {
            const double tl = tlo[k][i], tu = tup[k][i];
            pin[k][i] = ll[k][i] > tl ? -1 : (lu[k][i] > tu ? 1 : 0);
        }
        for (int i = 0; i < nu; i++) {
            if (!pin[k][i]) continue;
            const double v = pin[k][i] < 0 ? p->lo[k][i] : p->hi[k][i];
            for (int l = 0; l < NX; l++) { m.b[k][l] += p->B[k][l * nu + i] * v; m.q[k][l] += p->S[k][i * NX + l] * v; }
            for (int j = 0; j < nu; j++) if (!pin[k][j]) m.r[k][j] += p->R[k][j * nu + i] * v;
        }
        for (int i = 0; i < nu; i++) {
            if (!pin[k][i]) continue;
            const double v = pin[k][i] < 0 ? p->lo[k][i] : p->hi[k][i];
            for (int l = 0; l < NX; l++) { m.B[k][l * nu + i] = 0.0; m.S[k][i * NX + l] = 0.0; }
            for (int j = 0; j < nu; j++) if (j != i) { m.R[k][i * nu + j] = 0.0; m.R[k][j * nu + i] = 0.0; }
            m.r[k][i] = -m.R[k][i * nu + i] * v;
        }
    }
    int ok = 0;
    for (int pass = 0; pass < max_pass && !ok; pass++) {
        if (pass > 0) {   /* rebuild the pinned problem for the corrected active set */
            for (int k = 0; k < N; k++) {
                const int nu = p->nu[k];
                memcpy(m.B[k], p->B[k], sizeof(double) * NX * nu); memcpy(m.b[k], p->b[k], sizeof(double) * NX);
                memcpy(m.R[k], p->R[k], sizeof(double) * nu * nu); memcpy(m.S[k], p->S[k], sizeof(double) * nu * NX);
                memcpy(m.q[k], p->q[k], sizeof(double) * NX); memcpy(m.r[k], p->r[k], sizeof(double) * nu);
                for (int i = 0; i < nu; i++) {
                    if (!pin[k][i]) continue;
                    const double v = pin[k][i] < 0 ? p->lo[k][i] : p->hi[k][i];
                    for (int l = 0; l < NX; l++) { m.b[k][l] += p->B[k][l * nu + i] * v; m.q[k][l] += p->S[k][i * NX + l] * v; }
                    for (int j = 0; j < nu; j++) if (!pin[k][j]) m.r[k][j] += p->R[k][j * nu + i] * v;
                }
                for (int i = 0; i < nu; i++) {
                    if (!pin[k][i]) continue;
                    const double v = pin[k][i] < 0 ? p->lo[k][i] : p->hi[k][i];
                    for (int l = 0; l < NX; l++) { m.B[k][l * nu + i] = 0.0; m.S[k][i * NX + l] = 0.0; }
                    for (int j = 0; j < nu; j++) if (j != i) { m.R[k][i * nu + j] = 0.0; m.R[k][j * nu + i] = 0.0; }
                    m.r[k][i] = -m.R[k][i * nu + i] * v;
                    zero[k][i] = 0.0;
                }
            }
        }
        (*passes)++;
        double g = 0.0;
        if (riccati_backward(&m, zero, m.r, 0, 1, &fp, &g)) break;
        if (*gbase == 0.0) *gbase = g;
        if (*gbase > 0.0 && g / *gbase > *growth) *growth = g / *gbase;
        if (growth_max > 0.0 && g > growth_max * *gbase) { *untrusted = 1; break; }
        riccati_forward(&m, f, dx0, 0, uh, xh);
        double pi[NX];
        int changed = 0, nanf = 0;
        for (int k = N - 1; k >= 0; k--) {
            const int nu = p->nu[k];
            for (int i = 0; i < NX; i++) {          /* costate of the pinned problem at stage k + 1 */
                double s2 = fp.p[k + 1][i];
                for (int j = 0; j < NX; j++) s2 += fp.P[k + 1][i * NX + j] * xh[(k + 1) * NX + j];
                pi[i] = s2;
            }
            for (int i = 0; i < nu; i++)
                if (pin[k][i]) uh[k][i] = pin[k][i] < 0 ? p->lo[k][i] : p->hi[k][i];
            for (int i = 0; i < nu; i++) {
                const double lo = p->lo[k][i], hi = p->hi[k][i];
                if (pin[k][i]) {   /* multiplier sign */
                    double g = p->r[k][i];
                    for (int j = 0; j < nu; j++) g += p->R[k][i * nu + j] * uh[k][j];
                    for (int j = 0; j < NX; j++) g += p->S[k][i * NX + j] * xh[k * NX + j] + p->B[k][j * nu + i] * pi[j];
                    const double tol = 1e-9 * (1.0 + fabs(g));
                    gsave[k][i] = g;
                    if ((pin[k][i] < 0 && g < -tol) || (pin[k][i] > 0 && g > tol)) { newpin[k][i] = 0; changed = 1; }
                    else newpin[k][i] = pin[k][i];
                } else {           /* primal feasibility */
                    const double tol = 1e-9 * (1.0 + fabs(lo) + fabs(hi));
                    if (!(uh[k][i] == uh[k][i])) nanf = 1;
                    if (uh[k][i] < lo - tol) { newpin[k][i] = -1; changed = 1; }
                    else if (uh[k][i] > hi + tol) { newpin[k][i] = 1; changed = 1; }
                    else newpin[k][i] = 0;
                }
            }
        }
        for (int i = 0; i < (N + 1) * NX; i++) if (!(xh[i] == xh[i])) nanf = 1;
        if (nanf) break;
        if (getenv("ORC_POLISH_TRACE")) {   /* diagnostic (tools/dev/pin_trace.py): how the pin set moves from pass to pass */
            int npin = 0, add = 0, rel = 0, kmin = N, kmax = -1;
            for (int k = 0; k < N; k++)
                for (int i = 0; i < p->nu[k]; i++) {
                    npin += pin[k][i] != 0;
                    if (pin[k][i] == newpin[k][i]) continue;
                    if (k < kmin) kmin = k;
                    if (k > kmax) kmax = k;
                    if (newpin[k][i]) add++; else rel++;
                }
            fprintf(stderr, "pass %2d: %4d pinned, next pass pins %3d more and releases %3d, stages %d..%d\n", pass, npin, add, rel, kmin, kmax);
        }
        if (!changed) { ok = 1; break; }
        /* out of passes with a finished, finite pass whose pin set was not empty: the multipliers gsave / inputs uh of THIS pass seed the interior point */
        if (pass == max_pass - 1) {
            int anypin = 0;
            for (int k = 0; k < N; k++) for (int i = 0; i < p->nu[k]; i++) anypin |= pin[k][i] != 0;
            exhausted = anypin;
            break;                      /* (pin keeps the set the pass was solved with) */
        }
        for (int k = 0; k < N; k++) memcpy(pin[k], newpin[k], sizeof(int) * (size_t)p->nu[k]);
    }
    if (!ok && exhausted && warm) {
        /* The attempt ran out of passes (not: failed): its last pass is a near-solution with a near-correct active set.  The
         * interior-point iteration takes over FROM THERE instead of from its standard cold point: inputs of the last pass pushed
         * 1e-3 of the box width inside the bounds, multipliers = the pass's multiplier estimates of the pinned inputs, floored at
         * the central-path value of mu = 1e-3.  (Measured on config 5, N = 600: the interior point then needs ONE iteration to
         * reach the threshold of the next attempt, which ends within 5 passes - against 7-13 iterations from the cold point.)  */
        *warm = 1;
        for (int k = 0; k < N; k++)
            for (int i = 0; i < p->nu[k]; i++) {
                const double lo = p->lo[k][i], hi = p->hi[k][i], w = hi - lo;
                /* a pinned input keeps its multiplier estimate g and takes the slack mu / g that puts the pair ON the central
                 * path of mu (at most delta * w: a small g is floored instead); every other pair sits on it by construction */
                const double gl = pin[k][i] < 0 ? gsave[k][i] : 0.0, gh = pin[k][i] > 0 ? -gsave[k][i] : 0.0;
                double tl = ORC_WARM_DELTA * w, th = ORC_WARM_DELTA * w;
                if (gl * tl > ORC_WARM_MU) tl = ORC_WARM_MU / gl;
                if (gh * th > ORC_WARM_MU) th = ORC_WARM_MU / gh;
                double v = uh[k][i];
                if (v < lo + tl) v = lo + tl;
                if (v > hi - th) v = hi - th;
                u[k][i] = v;
                tlo[k][i] = v - lo;
                tup[k][i] = hi - v;
                ll[k][i] = ORC_WARM_MU / tlo[k][i];
                lu[k][i] = ORC_WARM_MU / tup[k][i];
            }
    }
    if (ok) {
        for (int k = 0; k < N; k++)
            for (int i = 0; i < p->nu[k]; i++) {
                u[k][i] = uh[k][i];
                ll[k][i] = pin[k][i] < 0 ? fmax(gsave[k][i], 0.0) : 0.0;
                lu[k][i] = pin[k][i] > 0 ? fmax(-gsave[k][i], 0.0) : 0.0;
            }
        memcpy(x, xh, sizeof(double) * (size_t)(N + 1) * NX);
    }
    for (int k = 0; k < N; k++) {
        orc_free(m.B[k]); orc_free(m.b[k]); orc_free(m.R[k]); orc_free(m.S[k]); orc_free(m.q[k]); orc_free(m.r[k]);
        orc_free(zero[k]); orc_free(uh[k]); orc_free(pin[k]); orc_free(newpin[k]); orc_free(gsave[k]);
    }
    for (int k = 0; k <= N; k++) { orc_free(fp.P[k]); orc_free(fp.p[k]); }
    orc_free(fp.P); orc_free(fp.p);
    orc_free(newpin); orc_free(gsave);
    orc_free(m.B); orc_free(m.b); orc_free(m.R); orc_free(m.S); orc_free(m.q); orc_free(m.r); orc_free(zero); orc_free(uh); orc_free(pin); orc_free(xh);
    return ok;
}

/* [UPSTREAM] HPIPM-style Mehrotra predictor-corrector interior point method on the
 * OCP-QP, Riccati factorisation of the KKT system, cold-started every call (U9).
 * The slacks of the input bounds are ITERATES of their own, as in HPIPM (d_ocp_qp_ipm: v, pi, lam, t): t_l, t_u start at
 * u - lo, hi - u and are updated t <- t + alpha dt with dt from the linearised bound equations, dt_l = du + (u - lo - t_l),
 * dt_u = -du + (hi - u - t_u); the bound residuals in the brackets (HPIPM's res_d) are zero to rounding - qp_bound_res = 1 feeds
 * them into the right-hand side of the Newton system as HPIPM does, the default leaves them out (measured: no difference).  A slack is never formed as the difference u - lo of two
 * numbers of magnitude 1-10 again: at mu <= 1e-11 with a multiplier in the thousands the central path puts it at 1e-14,
 * below the resolution of that difference (round 4: 13 fuzz instances ended NaN here because u - lo rounded to 0).
 * States are implied by the (affine) dynamics, so the residuals that matter are stationarity
 * (scales by 1-alpha per step; tracked as rho) and complementarity (mu).                */
/* states of the QP from its inputs (the dynamics are affine): x_0 = dx0, x_{k+1} = A x + B u + b */
static void ipm_rollout(const ocpqp *p, const double *dx0, double **u, double *x)
{
    const int N = p->N;
    for (int i = 0; i < NX; i++) x[i] = dx0 ? dx0[i] : 0.0;
    for (int k = 0; k < N; k++)
        for (int i = 0; i < NX; i++) {
            double s = p->b[k][i];
            for (int j = 0; j < NX; j++) s += p->A[k][i * NX + j] * x[k * NX + j];
            for (int j = 0; j < p->nu[k]; j++) s += p->B[k][i * p->nu[k] + j] * u[k][j];
            x[(k + 1) * NX + i] = s;
        }
}

/* TRUE residuals of a primal-dual point, recomputed from scratch in the infinity norm: stationarity of the inputs (costates by
 * the adjoint recursion from the rolled-out states), complementarity max lam * slack, bound equations max |u - lo - t_l|,
 * |hi - u - t_u|.  tlo / tup NULL: slacks taken as u - lo, hi - u (an accepted active-set solution carries none).  xin NULL:
 * the states are rolled out here (x is then scratch of (N+1)*NX doubles allocated inside).                                   */
static void ipm_true_residuals(const ocpqp *p, const double *dx0, double **u, double **ll, double **lu, double **tlo, double **tup,
                               const double *xin, double *res_stat, double *res_comp, double *res_bnd)
{
    const int N = p->N;
    double *xs = NULL;
    if (!xin) { xs = dalloc((size_t)(N + 1) * NX); ipm_rollout(p, dx0, u, xs); }
    const double *x = xin ? xin : xs;
    double pi[NX], pin[NX], rs = 0.0, rc = 0.0, rb = 0.0;
    for (int i = 0; i < NX; i++) {
        double s = p->qN[i];
        for (int j = 0; j < NX; j++) s += p->QN[i * NX + j] * x[N * NX + j];
        pi[i] = s;
    }
    for (int k = N - 1; k >= 0; k--) {
        const int m = p->nu[k];
        for (int i = 0; i < m; i++) {
            double s = p->r[k][i] - ll[k][i] + lu[k][i];
            for (int j = 0; j < m; j++) s += p->R[k][i * m + j] * u[k][j];
            for (int j = 0; j < NX; j++) s += p->S[k][i * NX + j] * x[k * NX + j];
            for (int j = 0; j < NX; j++) s += p->B[k][j * m + i] * pi[j];
            if (fabs(s) > rs) rs = fabs(s);
            const double sl = u[k][i] - p->lo[k][i], su = p->hi[k][i] - u[k][i];
            const double tl = tlo ? tlo[k][i] : sl, tu = tup ? tup[k][i] : su;
            const double c1 = ll[k][i] * tl, c2 = lu[k][i] * tu;
            if (c1 > rc) rc = c1;
            if (c2 > rc) rc = c2;
            if (fabs(sl - tl) > rb) rb = fabs(sl - tl);
            if (fabs(su - tu) > rb) rb = fabs(su - tu);
        }
        for (int i = 0; i < NX; i++) {
            double s = p->q[k][i];
            for (int j = 0; j < NX; j++) s += p->Q[k][i * NX + j] * x[k * NX + j];
            for (int j = 0; j < m; j++) s += p->S[k][j * NX + i] * u[k][j];
            for (int j = 0; j < NX; j++) s += p->A[k][j * NX + i] * pi[j];
            pin[i] = s;
        }
        memcpy(pi, pin, sizeof(pi));
    }
    if (xs) orc_free(xs);
    *res_stat = rs; *res_comp = rc; *res_bnd = rb;
}

/* attempt policy of the active-set passes where the configuration leaves it at 0: the library's rule (csrc/nmpc_consts.hpp,
 * resolve_polish_policy, which has the measurements) - 8 passes per attempt and 16 in total below N = 160, ONE attempt of N / 16 passes
 * (16 .. 32) from there up */
static void orc_polish_policy(const orc_config *c, int N, int *passes, int *budget)
{
    int lp = N / 16;                       /* one attempt of N / 16 passes, at least 16, at most 32, from N = 160 up */
    lp = lp < 16 ? 16 : (lp > 32 ? 32 : lp);
    *passes = c->qp_polish_passes > 0 ? c->qp_polish_passes : (N >= 160 ? lp : 8);
    *budget = c->qp_polish_budget > 0 ? c->qp_polish_budget : (N >= 160 ? (*passes > 16 ? *passes : 16) : 2 * *passes);
}

/* standard (cold) start point of the interior-point iteration: inputs pushed inside the box, multipliers on the central path of mu0 */
static void ipm_cold_point(const orc_config *c, const ocpqp *p, double **u, double **ll, double **lu, double **tlo, double **tup)
{
    for (int k = 0; k < p->N; k++)
        for (int i = 0; i < p->nu[k]; i++) {
            const double lo = p->lo[k][i], hi = p->hi[k][i];
            double thr = c->qp_thr0;
            if (c->qp_thr0_rel * (hi - lo) > thr) thr = c->qp_thr0_rel * (hi - lo);
            if (hi - lo < 2.0 * thr) thr = 0.5 * (hi - lo);
            double v = 0.0;
            if (v - lo < thr) v = lo + thr;
            if (hi - v < thr) v = hi - thr;
            u[k][i] = v;
            tlo[k][i] = v - lo;
            tup[k][i] = hi - v;
            ll[k][i] = c->qp_mu0 / tlo[k][i];
            lu[k][i] = c->qp_mu0 / tup[k][i];
        }
}

static int ocpqp_ipm(const orc_config *c, const ocpqp *p, const double *dx0,
                     double **u /*out*/, double *x /*out (N+1)*NX*/, orc_stats *st)
{
    const int N = p->N;
    int nc = 0, status = 0, it = 0;
    ricc_fact f;
    f.L = (double **)orc_malloc(sizeof(double *) * N);
    f.M = (double **)orc_malloc(sizeof(double *) * N);
    f.m = (double **)orc_malloc(sizeof(double *) * N);
    f.P = NULL; f.p = NULL;
    double **ll = (double **)orc_malloc(sizeof(double *) * N), **lu = (double **)orc_malloc(sizeof(double *) * N);
    double **sig = (double **)orc_malloc(sizeof(double *) * N), **rh = (double **)orc_malloc(sizeof(double *) * N);
    double **ua = (double **)orc_malloc(sizeof(double *) * N), **du = (double **)orc_malloc(sizeof(double *) * N);
    double **dla = (double **)orc_malloc(sizeof(double *) * N), **dua = (double **)orc_malloc(sizeof(double *) * N);
    /* slacks of the input bounds (iterates), their directions, and the bound residuals of the iteration in flight */
    double **tlo = (double **)orc_malloc(sizeof(double *) * N), **tup = (double **)orc_malloc(sizeof(double *) * N);
    double **dtl = (double **)orc_malloc(sizeof(double *) * N), **dtu = (double **)orc_malloc(sizeof(double *) * N);
    double **rbl = (double **)orc_malloc(sizeof(double *) * N), **rbu = (double **)orc_malloc(sizeof(double *) * N);
    double *xh = dalloc((size_t)(N + 1) * NX);
    double **bsh = (double **)orc_malloc(sizeof(double *) * N);     /* b_k + B_k u_k of the iterate (input-delta form of the Newton system) */
    double **qsh = (double **)orc_malloc(sizeof(double *) * N);     /* q_k + S_k' u_k (the cross term of a condensed stage) */
    ocpqp ps = *p;
    ps.b = bsh; ps.q = qsh;
    for (int k = 0; k < N; k++) {
        const int m = p->nu[k];
        nc += 2 * m;
        bsh[k] = dalloc(NX); qsh[k] = dalloc(NX);
        f.L[k] = dalloc((size_t)m * m); f.M[k] = dalloc((size_t)m * NX); f.m[k] = dalloc(m);
        ll[k] = dalloc(m); lu[k] = dalloc(m); sig[k] = dalloc(m); rh[k] = dalloc(m);
        ua[k] = dalloc(m); du[k] = dalloc(m); dla[k] = dalloc(m); dua[k] = dalloc(m);
        tlo[k] = dalloc(m); tup[k] = dalloc(m); dtl[k] = dalloc(m); dtu[k] = dalloc(m); rbl[k] = dalloc(m); rbu[k] = dalloc(m);
    }
    ipm_cold_point(c, p, u, ll, lu, tlo, tup);
    double rho = 1.0, mu = 0.0, pol_mu = c->qp_polish_mu;
    double gbase = 0.0, growth = 0.0, step_last = 0.0;
    int polished = 0, npolish = 0, untrusted = 0;
    int warm_derived = 0;   /* the iterate descends from a warm start (an exhausted attempt's last pass), not from the cold point */
    const int itmax = c->qp_iter_max > 0 ? c->qp_iter_max : 1;
    int pol_passes, pol_budget;
    orc_polish_policy(c, c->N, &pol_passes, &pol_budget);
    /* [UPSTREAM U9] exit of HPIPM at its DEFAULT tolerances (qp_exit_mode = 1; acados leaves res_g/b/d/m_max at 1e-8 and
     * controller.py:179-190 sets none): all four TRUE residuals of the iterate - stationarity, dynamics (0 here: the states
     * are implied), bounds (|u - lo - t|), complementarity (max lam t, as HPIPM's res_m with mu = 0) - at most qp_tol_stat /
     * qp_tol_comp in the infinity norm.  Used to PREDICT acados' own accuracy floor (tests/test_acados_floor.py), not by
     * the shipped defaults (mode 0: mean complementarity + stationarity factor, far tighter tolerances).                */
    for (;;) {
        mu = 0.0;
        for (int k = 0; k < N; k++)
            for (int i = 0; i < p->nu[k]; i++)
                mu += ll[k][i] * tlo[k][i] + lu[k][i] * tup[k][i];
        mu /= nc;
        if (!(mu == mu)) { status = 1; break; }
        if (c->qp_exit_mode == 1) {
            double rs = 0.0, rc = 0.0, rb = 0.0;
            ipm_true_residuals(p, dx0, u, ll, lu, tlo, tup, NULL, &rs, &rc, &rb);
            if (rs <= c->qp_tol_stat && rc <= c->qp_tol_comp && rb <= c->qp_tol_stat) break;
        } else if (mu <= c->qp_tol_comp && rho <= c->qp_tol_stat && (it == 0 || !(c->qp_tol_step > 0.0) || step_last <= c->qp_tol_step)) break;
        if (c->qp_polish && mu <= pol_mu && npolish < pol_budget) {
            int trip = 0, warm = 0;
            /* the warm start is for the iteration BETWEEN two attempts: an attempt that uses up the budget leaves the iterate alone */
            const int last_attempt = npolish + pol_passes >= pol_budget;
            if (ocpqp_polish(p, dx0, &f, u, ll, lu, tlo, tup, x, pol_passes, &npolish, c->qp_growth_max, &gbase, &growth, &trip,
                             (c->qp_warm_start && !last_attempt) ? &warm : NULL)) {
                polished = 1; mu = 0.0; rho = 0.0; break;
            }
            if (trip) { untrusted = 1; npolish = pol_budget; }     /* the same pins would fail the same way: no further attempt */
            pol_mu *= 1e-2;
            if (warm) {      /* the iterate was replaced: its duality measure, and nothing known about its stationarity */
                mu = 0.0;
                for (int k = 0; k < N; k++)
                    for (int i = 0; i < p->nu[k]; i++)
                        mu += ll[k][i] * tlo[k][i] + lu[k][i] * tup[k][i];
                mu /= nc;
                rho = 1.0;
                warm_derived = 1;
                /* (pol_mu is not touched: the warm point's mu ~ 1e-3 is already below the threshold of the next attempt, which
                 * therefore follows after ONE interior-point iteration - the iteration that re-derives the active-set guess) */
            } else if (warm_derived && npolish >= pol_budget) {
                /* no attempt is left and the iterate in hand descends from a warm start - off the central path, a poor place to
                 * converge from (42 iterations and 1e-6 of accuracy on fuzz draw 353, against 15 and 2e-8): the interior point
                 * finishes the QP from its standard cold point                                                                */
                ipm_cold_point(c, p, u, ll, lu, tlo, tup);
                mu = c->qp_mu0;
                rho = 1.0;
                warm_derived = 0;
            }
        }
        if (it >= itmax) { status = 2; break; }
        it++;
        /* predictor (affine scaling), solved for the STEP of the inputs: with u = u_it + w the stage reads x+ = A x + B w +
         * (b + B u_it), input gradient r + R u_it, state gradient q + S'u_it, input Hessian R + sig - states stay absolute, inputs become deltas.  The
         * target form of rounds 1-4 (solve for ua, then d = ua - u) loses everything of d below ulp(u), and a slack carried to
         * 1e-14 needs its direction to that accuracy: the step of a nearly active input is d ~ -t, formed here as
         * gradient / (R + sig) without a cancellation.  Bound residuals of the iterate (HPIPM's res_d): with qp_bound_res = 1
         * they stay fixed over the iteration and enter the affine right-hand side, so that the step restores u - lo = t_l,
         * hi - u = t_u as far as u can resolve it; the start is feasible and t follows u exactly in exact arithmetic, so they hold
         * rounding of size ulp(u) and the default (0, what the kernels do) leaves them out - tests/test_oracle_qp.py shows
         * iteration counts and commands unchanged either way */
        for (int k = 0; k < N; k++) {
            const int m = p->nu[k];
            for (int i = 0; i < NX; i++) {
                double s2 = p->b[k][i];
                for (int j = 0; j < m; j++) s2 += p->B[k][i * m + j] * u[k][j];
                bsh[k][i] = s2;
                double s3 = p->q[k][i];
                for (int j = 0; j < m; j++) s3 += p->S[k][j * NX + i] * u[k][j];
                qsh[k][i] = s3;
            }
            for (int i = 0; i < m; i++) {
                const double tl = tlo[k][i], tu = tup[k][i];
                rbl[k][i] = c->qp_bound_res ? (u[k][i] - p->lo[k][i]) - tl : 0.0;
                rbu[k][i] = c->qp_bound_res ? (p->hi[k][i] - u[k][i]) - tu : 0.0;
                sig[k][i] = ll[k][i] / tl + lu[k][i] / tu;
                double s2 = p->r[k][i];
                for (int j = 0; j < m; j++) s2 += p->R[k][i * m + j] * u[k][j];
                rh[k][i] = s2 + ll[k][i] / tl * rbl[k][i] - lu[k][i] / tu * rbu[k][i];
            }
        }
        {
            double g = 0.0;
            const int fail = riccati_backward(&ps, sig, rh, 0, 1, &f, &g);
            if (!fail && gbase == 0.0) gbase = g;
            if (!fail && gbase > 0.0 && g / gbase > growth) growth = g / gbase;
            const int trip = !fail && c->qp_growth_max > 0.0 && !(g <= c->qp_growth_max * gbase);
            if (fail || trip) {
                /* this factorisation cannot be used: the QP ends at the current iterate - solved if that is within the
                 * acceptable tolerances, a QP failure otherwise */
                if (trip) untrusted = 1;
                status = (mu <= c->qp_acc_comp && rho <= c->qp_acc_stat) ? 0 : (trip ? 4 : (fail == 2 ? 1 : 3));
                break;
            }
        }
        riccati_forward(&ps, &f, dx0, 0, ua, xh);      /* ua: the affine STEP of the inputs */
        double aaff = 1.0;
        for (int k = 0; k < N; k++)
            for (int i = 0; i < p->nu[k]; i++) {
                const double tl = tlo[k][i], tu = tup[k][i];
                const double d = ua[k][i];
                const double el = d + rbl[k][i], eu = -d + rbu[k][i];      /* affine directions of the two slacks */
                dla[k][i] = -ll[k][i] - ll[k][i] / tl * el;
                dua[k][i] = -lu[k][i] - lu[k][i] / tu * eu;
                if (el < 0.0 && -tl / el < aaff) aaff = -tl / el;
                if (eu < 0.0 && -tu / eu < aaff) aaff = -tu / eu;
                if (dla[k][i] < 0.0 && -ll[k][i] / dla[k][i] < aaff) aaff = -ll[k][i] / dla[k][i];
                if (dua[k][i] < 0.0 && -lu[k][i] / dua[k][i] < aaff) aaff = -lu[k][i] / dua[k][i];
            }
        double muaff = 0.0;
        for (int k = 0; k < N; k++)
            for (int i = 0; i < p->nu[k]; i++) {
                const double tl = tlo[k][i], tu = tup[k][i];
                const double d = ua[k][i];
                const double el = d + rbl[k][i], eu = -d + rbu[k][i];
                muaff += (ll[k][i] + aaff * dla[k][i]) * (tl + aaff * el) +
                         (lu[k][i] + aaff * dua[k][i]) * (tu + aaff * eu);
            }
        muaff /= nc;
        double sg = muaff / mu;
        sg = sg * sg * sg;
        /* corrector: homogeneous solve for the change of the input gradient */
        for (int k = 0; k < N; k++)
            for (int i = 0; i < p->nu[k]; i++) {
                const double tl = tlo[k][i], tu = tup[k][i];
                const double d = ua[k][i];
                const double el = d + rbl[k][i], eu = -d + rbu[k][i];
                const double cl = dla[k][i] * el, cu = dua[k][i] * eu;
                rh[k][i] = -(sg * mu - cl) / tl + (sg * mu - cu) / tu;
            }
        riccati_backward(p, sig, rh, 1, 0, &f, NULL);
        riccati_forward(p, &f, NULL, 1, du, xh);
        double amax = 1e300;
        for (int k = 0; k < N; k++)
            for (int i = 0; i < p->nu[k]; i++) {
                const double tl = tlo[k][i], tu = tup[k][i];
                const double da = ua[k][i];
                const double el = da + rbl[k][i], eu = -da + rbu[k][i];
                const double cl = dla[k][i] * el, cu = dua[k][i] * eu;
                const double d = da + du[k][i];
                du[k][i] = d;
                dtl[k][i] = d + rbl[k][i];
                dtu[k][i] = -d + rbu[k][i];
                dla[k][i] = -(ll[k][i] * tl + cl - sg * mu) / tl - ll[k][i] / tl * dtl[k][i];
                dua[k][i] = -(lu[k][i] * tu + cu - sg * mu) / tu - lu[k][i] / tu * dtu[k][i];
                if (dtl[k][i] < 0.0 && -tl / dtl[k][i] < amax) amax = -tl / dtl[k][i];
                if (dtu[k][i] < 0.0 && -tu / dtu[k][i] < amax) amax = -tu / dtu[k][i];
                if (dla[k][i] < 0.0 && -ll[k][i] / dla[k][i] < amax) amax = -ll[k][i] / dla[k][i];
                if (dua[k][i] < 0.0 && -lu[k][i] / dua[k][i] < amax) amax = -lu[k][i] / dua[k][i];
            }
        double alpha = c->qp_tau * amax;
        if (alpha > 1.0) alpha = 1.0;
        if (!(alpha == alpha)) { status = 1; break; }
        /* centrality safeguard (wide neighbourhood N_-inf(gamma)): shorten the step until
         * no complementarity product falls below gamma * (new mean).  Without it the
         * plain Mehrotra heuristic can 2-cycle when one pair blocks the affine step.     */
        if (c->qp_gamma > 0.0) {
            for (int bt = 0; bt < 30; bt++) {
                double mun = 0.0, pmin = 1e300;
                for (int k = 0; k < N; k++)
                    for (int i = 0; i < p->nu[k]; i++) {
                        const double p1 = (ll[k][i] + alpha * dla[k][i]) * (tlo[k][i] + alpha * dtl[k][i]);
                        const double p2 = (lu[k][i] + alpha * dua[k][i]) * (tup[k][i] + alpha * dtu[k][i]);
                        mun += p1 + p2;
                        if (p1 < pmin) pmin = p1;
                        if (p2 < pmin) pmin = p2;
                    }
                mun /= nc;
                if (pmin >= c->qp_gamma * mun) break;
                alpha *= 0.75;
            }
        }
        if (alpha < 1e-12) { status = 3; break; }
        step_last = 0.0;
        for (int k = 0; k < N; k++)
            for (int i = 0; i < p->nu[k]; i++) {
                const double sw = fabs(alpha * du[k][i]) / (p->hi[k][i] - p->lo[k][i]);
                if (sw > step_last) step_last = sw;
                u[k][i] += alpha * du[k][i];
                tlo[k][i] += alpha * dtl[k][i];
                tup[k][i] += alpha * dtu[k][i];
                ll[k][i] += alpha * dla[k][i];
                lu[k][i] += alpha * dua[k][i];
            }
        rho *= (1.0 - alpha);
        if (getenv("ORC_DEBUG")) fprintf(stderr, "it %d mu %.3e aaff %.3e muaff %.3e sg %.3e alpha %.3e rho %.3e\n", it, mu, aaff, muaff, sg, alpha, rho);
    }
    /* final rollout of the states from the inputs (dynamics are affine) */
    if (!polished) ipm_rollout(p, dx0, u, x);
    if (st) {
        /* TRUE residuals, recomputed from scratch (diagnostics for the tests) */
        double rs = 0.0, rc = 0.0, rb = 0.0;
        ipm_true_residuals(p, dx0, u, ll, lu, polished ? NULL : tlo, polished ? NULL : tup, polished ? x : NULL, &rs, &rc, &rb);
        st->qp_iter = it; st->qp_status = status; st->res_stat = rs; st->res_eq = 0.0;
        st->res_comp = rc; st->mu = mu; st->rho = rho; st->polished = polished; st->polish_attempts = npolish;
        st->growth = growth; st->step_last = step_last; st->untrusted = untrusted;
    }
    for (int k = 0; k < N; k++) {
        orc_free(f.L[k]); orc_free(f.M[k]); orc_free(f.m[k]); orc_free(ll[k]); orc_free(lu[k]); orc_free(sig[k]);
        orc_free(rh[k]); orc_free(ua[k]); orc_free(du[k]); orc_free(dla[k]); orc_free(dua[k]);
        orc_free(tlo[k]); orc_free(tup[k]); orc_free(dtl[k]); orc_free(dtu[k]); orc_free(rbl[k]); orc_free(rbu[k]); orc_free(bsh[k]); orc_free(qsh[k]);
    }
    orc_free(f.L); orc_free(f.M); orc_free(f.m); orc_free(ll); orc_free(lu); orc_free(sig); orc_free(rh); orc_free(ua);
    orc_free(du); orc_free(dla); orc_free(dua); orc_free(tlo); orc_free(tup); orc_free(dtl); orc_free(dtu); orc_free(rbl); orc_free(rbu);
    orc_free(xh); orc_free(bsh); orc_free(qsh);
    return status;
}

/* [UPSTREAM] HPIPM partial condensing block sizes: N stages into N2 blocks, remainder
 * spread over the first blocks.                                                          */
static void cond_block_sizes(int N, int N2, int *bs)
{
    const int base = N / N2, rem = N - N2 * base;
    for (int i = 0; i < N2; i++) bs[i] = base + (i < rem ? 1 : 0);
}

int orc_qp_solve(const orc_config *c, const double *dx0,
                 const double *A, const double *B, const double *b,
                 const double *q, const double *r, const double *lo, const double *hi,
                 const double *Qd, const double *Rd,
                 double *dx, double *du, orc_stats *st)
{
    const int N = c->N;
    int N2 = (c->qp_cond_N > 0 && c->qp_cond_N < N) ? c->qp_cond_N : N;
    int *bs = (int *)orc_malloc(sizeof(int) * (size_t)N2);
    int *nu2 = (int *)orc_malloc(sizeof(int) * (size_t)N2);
    cond_block_sizes(N, N2, bs);
    for (int i = 0; i < N2; i++) nu2[i] = NU * bs[i];
    ocpqp p;
    ocpqp_alloc(&p, N2, nu2);
    /* per block: Phi_j (nx*nx), Gam_j (nx*nub), c_j (nx) for j = 0..bs (U8) */
    int k0 = 0;
    double **Phi_all = (double **)orc_malloc(sizeof(double *) * N2);
    double **Gam_all = (double **)orc_malloc(sizeof(double *) * N2);
    double **c_all = (double **)orc_malloc(sizeof(double *) * N2);
    for (int ib = 0; ib < N2; ib++) {
        const int nb = bs[ib], m = nu2[ib];
        double *Phi = dalloc((size_t)(nb + 1) * NX * NX);
        double *Gam = dalloc((size_t)(nb + 1) * NX * m);
        double *cc = dalloc((size_t)(nb + 1) * NX);
        Phi_all[ib] = Phi; Gam_all[ib] = Gam; c_all[ib] = cc;
        for (int i = 0; i < NX; i++) Phi[i * NX + i] = 1.0;
        for (int j = 0; j < nb; j++) {
            const double *Ak = A + (size_t)(k0 + j) * NX * NX, *Bk = B + (size_t)(k0 + j) * NX * NU;
            const double *bk = b + (size_t)(k0 + j) * NX;
            const double *Pj = Phi + (size_t)j * NX * NX, *Gj = Gam + (size_t)j * NX * m, *cj = cc + (size_t)j * NX;
            double *Pn = Phi + (size_t)(j + 1) * NX * NX, *Gn = Gam + (size_t)(j + 1) * NX * m, *cn = cc + (size_t)(j + 1) * NX;
            for (int i = 0; i < NX; i++) {
                for (int l = 0; l < NX; l++) {
                    double s = 0.0;
                    for (int t = 0; t < NX; t++) s += Ak[i * NX + t] * Pj[t * NX + l];
                    Pn[i * NX + l] = s;
                }
                for (int l = 0; l < m; l++) {
                    double s = 0.0;
                    for (int t = 0; t < NX; t++) s += Ak[i * NX + t] * Gj[t * m + l];
                    Gn[i * m + l] = s;
                }
                for (int l = 0; l < NU; l++) Gn[i * m + j * NU + l] += Bk[i * NU + l];
                double s = bk[i];
                for (int t = 0; t < NX; t++) s += Ak[i * NX + t] * cj[t];
                cn[i] = s;
            }
        }
        memcpy(p.A[ib], Phi + (size_t)nb * NX * NX, sizeof(double) * NX * NX);
        memcpy(p.B[ib], Gam + (size_t)nb * NX * m, sizeof(double) * NX * m);
        memcpy(p.b[ib], cc + (size_t)nb * NX, sizeof(double) * NX);
        for (int j = 0; j < nb; j++) {
            const double *Pj = Phi + (size_t)j * NX * NX, *Gj = Gam + (size_t)j * NX * m, *cj = cc + (size_t)j * NX;
            const double *Qk = Qd + (size_t)(k0 + j) * NX, *qk = q + (size_t)(k0 + j) * NX;
            double w[NX];
            for (int t = 0; t < NX; t++) w[t] = Qk[t] * cj[t] + qk[t];
            for (int i = 0; i < NX; i++) {
                for (int l = 0; l < NX; l++) {
                    double s = 0.0;
                    for (int t = 0; t < NX; t++) s += Pj[t * NX + i] * Qk[t] * Pj[t * NX + l];
                    p.Q[ib][i * NX + l] += s;
                }
                double s = 0.0;
                for (int t = 0; t < NX; t++) s += Pj[t * NX + i] * w[t];
                p.q[ib][i] += s;
            }
            for (int i = 0; i < m; i++) {
                for (int l = 0; l < NX; l++) {
                    double s = 0.0;
                    for (int t = 0; t < NX; t++) s += Gj[t * m + i] * Qk[t] * Pj[t * NX + l];
                    p.S[ib][i * NX + l] += s;
                }
                for (int l = 0; l < m; l++) {
                    double s = 0.0;
                    for (int t = 0; t < NX; t++) s += Gj[t * m + i] * Qk[t] * Gj[t * m + l];
                    p.R[ib][i * m + l] += s;
                }
                double s = 0.0;
                for (int t = 0; t < NX; t++) s += Gj[t * m + i] * w[t];
                p.r[ib][i] += s;
            }
            for (int l = 0; l < NU; l++) {
                const int i = j * NU + l;
                p.R[ib][i * m + i] += Rd[(size_t)(k0 + j) * NU + l];
                p.r[ib][i] += r[(size_t)(k0 + j) * NU + l];
                p.lo[ib][i] = lo[(size_t)(k0 + j) * NU + l];
                p.hi[ib][i] = hi[(size_t)(k0 + j) * NU + l];
            }
        }
        k0 += nb;
    }
    for (int i = 0; i < NX; i++) {
        p.QN[i * NX + i] = Qd[(size_t)N * NX + i];
        p.qN[i] = q[(size_t)N * NX + i];
    }
    double **u2 = (double **)orc_malloc(sizeof(double *) * N2);
    for (int ib = 0; ib < N2; ib++) u2[ib] = dalloc(nu2[ib]);
    double *x2 = dalloc((size_t)(N2 + 1) * NX);
    const int status = ocpqp_ipm(c, &p, dx0, u2, x2, st);
    /* expansion back to the N-stage trajectory */
    k0 = 0;
    for (int ib = 0; ib < N2; ib++) {
        const int nb = bs[ib], m = nu2[ib];
        for (int j = 0; j < nb; j++) {
            for (int l = 0; l < NU; l++) du[(size_t)(k0 + j) * NU + l] = u2[ib][j * NU + l];
            const double *Pj = Phi_all[ib] + (size_t)j * NX * NX, *Gj = Gam_all[ib] + (size_t)j * NX * m;
            const double *cj = c_all[ib] + (size_t)j * NX;
            for (int i = 0; i < NX; i++) {
                double s = cj[i];
                for (int t = 0; t < NX; t++) s += Pj[i * NX + t] * x2[ib * NX + t];
                for (int t = 0; t < m; t++) s += Gj[i * m + t] * u2[ib][t];
                dx[(size_t)(k0 + j) * NX + i] = s;
            }
        }
        k0 += nb;
    }
    memcpy(dx + (size_t)N * NX, x2 + (size_t)N2 * NX, sizeof(double) * NX);
    for (int ib = 0; ib < N2; ib++) {
        orc_free(u2[ib]); orc_free(Phi_all[ib]); orc_free(Gam_all[ib]); orc_free(c_all[ib]);
    }
    orc_free(u2); orc_free(x2); orc_free(Phi_all); orc_free(Gam_all); orc_free(c_all); orc_free(bs); orc_free(nu2);
    ocpqp_free(&p);
    return status;
}

/* ------------------------------------------------------------------------------------ */
/* [UPSTREAM] ocp_nlp_sqp_rti: preparation + feedback, full step (U1), status (U10).      */
int orc_sqp_rti(const orc_config *c, const double *x0, const double *yref,
                const double *yref_e, double *xtraj, double *utraj, orc_stats *st)
{
    const int N = c->N;
    double *A = dalloc((size_t)N * NX * NX), *B = dalloc((size_t)N * NX * NU), *b = dalloc((size_t)N * NX);
    double *q = dalloc((size_t)(N + 1) * NX), *r = dalloc((size_t)N * NU);
    double *lo = dalloc((size_t)N * NU), *hi = dalloc((size_t)N * NU);
    double *Qd = dalloc((size_t)(N + 1) * NX), *Rd = dalloc((size_t)N * NU);
    double *dx = dalloc((size_t)(N + 1) * NX), *du = dalloc((size_t)N * NU);
    double dx0[NX];
    int projected = 0, status = 0;
    orc_stats local;
    memset(&local, 0, sizeof(local));
    /* what the caller handed in: finite or not (decides the class of a failure below) */
    int in_nf = 0;
    for (int i = 0; i < NX; i++) in_nf |= !(fabs(x0[i]) <= DBL_MAX) || !(fabs(yref_e[i]) <= DBL_MAX);
    for (int i = 0; i < N * NY; i++) in_nf |= !(fabs(yref[i]) <= DBL_MAX);
    for (int i = 0; i < (N + 1) * NX; i++) in_nf |= !(fabs(xtraj[i]) <= DBL_MAX);
    for (int i = 0; i < N * NU; i++) in_nf |= !(fabs(utraj[i]) <= DBL_MAX);
    orc_linearize(c, xtraj, utraj, yref, yref_e, A, B, b, q, r, lo, hi, Qd, Rd, &projected);
    /* lbx_0 = ubx_0 = x0 (controller.py:414-415): delta x_0 is pinned to x0 - x_0 (U7) */
    for (int i = 0; i < NX; i++) dx0[i] = x0[i] - xtraj[i];
    const int qps = orc_qp_solve(c, dx0, A, B, b, q, r, lo, hi, Qd, Rd, dx, du, &local);
    local.hess_projected = projected;
    int bad = 0;
    for (int i = 0; i < (N + 1) * NX; i++) if (!(dx[i] == dx[i]) || fabs(dx[i]) > 1e300) bad = 1;
    for (int i = 0; i < N * NU; i++) if (!(du[i] == du[i]) || fabs(du[i]) > 1e300) bad = 1;
    /* (a QP that failed is a QP failure whatever the step it left holds - acados returns ACADOS_QP_FAILURE before it looks at the step;
     * NaN data shows as a NaN pivot inside the QP: qps == 1.  Until round 5 a NaN anywhere in the discarded step of a failed QP read as
     * status 1: seed 431 of the fuzz, a warm start at |x| 8e5, came back 1 here and 4 from the kernels) */
    if (qps == 1) status = 1;                  /* ACADOS_NAN_DETECTED */
    else if (qps == 3 || qps == 4) status = 4; /* QP min step / factorisation / untrusted factorisation -> QP_FAILURE */
    else if (bad) status = 1;
    else if (qps == 2) status = c->qp_maxiter_status ? 2 : 0;   /* QP max-iter: tolerated in RTI or reported (U10 switch) */
    else status = 0;
    /* The CLASS of a failure follows the inputs, not the arithmetic (late round 5): ACADOS_NAN_DETECTED (1) exactly when a value the caller
     * handed in - x0, yref, yref_e, the linearisation trajectory - is not finite, ACADOS_QP_FAILURE (4) for every other failed solve.  On
     * data that have left the model's range (warm starts about diverged trajectories, |x| 5e3 .. 5e10: pivots of 1e116 .. 1e269) whether a
     * pivot came out NaN, out of range or merely not positive was decided by which sum overflowed or cancelled first - fuzz draws 431 and
     * 11856 ended 1 on one side and 4 on the other, either way round.  The reference treats every non-zero status alike
     * (controller.py:448-450); the kernels class their failures the same way (csrc/nmpc_ipm.hpp inputs_not_finite). */
    if (status == 1 || status == 4) status = in_nf ? 1 : 4;
    if (status == 0) {
        for (int i = 0; i < (N + 1) * NX; i++) xtraj[i] += dx[i];
        for (int i = 0; i < N * NU; i++) utraj[i] += du[i];
    }
    if (st) *st = local;
    orc_free(A); orc_free(B); orc_free(b); orc_free(q); orc_free(r); orc_free(lo); orc_free(hi); orc_free(Qd); orc_free(Rd);
    orc_free(dx); orc_free(du);
    return status;
}

int orc_solve_batch(const orc_config *c, int Bn, const double *x0, const double *yref,
                    const double *yref_e, int yref_bcast,
                    const double *x_init, const double *u_init,
                    double *u0, int *status, double *x_out, double *u_out,
                    int *iters, int nthreads)
{
    return orc_solve_batch_ex(c, Bn, x0, yref, yref_e, yref_bcast, x_init, u_init, u0, status, x_out, u_out, iters, NULL, NULL, nthreads);
}

int orc_solve_batch_ex(const orc_config *c, int Bn, const double *x0, const double *yref,
                       const double *yref_e, int yref_bcast,
                       const double *x_init, const double *u_init,
                       double *u0, int *status, double *x_out, double *u_out,
                       int *iters, int *passes, double *growth, int nthreads)
{
    const int N = c->N;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel
    {
        /* one arena per thread, reset per instance, grown between instances to what the largest one asked for */
        orc_arena ar = {NULL, 0, 0, 0};
        if (ORC_USE_ARENA) {
            ar.cap = (size_t)1 << 20;
            ar.base = (char *)malloc(ar.cap);
            if (!ar.base) ar.cap = 0;
            t_arena = &ar;
        }
#pragma omp for schedule(static)
    for (int ib = 0; ib < Bn; ib++) {
        ar.off = 0; ar.need = 0;
        double *xt = (double *)orc_malloc(sizeof(double) * (size_t)(N + 1) * NX);
        double *ut = (double *)orc_malloc(sizeof(double) * (size_t)N * NU);
        const double *x0i = x0 + (size_t)ib * NX;
        if (x_init && u_init) {
            memcpy(xt, x_init + (size_t)ib * (N + 1) * NX, sizeof(double) * (size_t)(N + 1) * NX);
            memcpy(ut, u_init + (size_t)ib * N * NU, sizeof(double) * (size_t)N * NU);
            memcpy(xt, x0i, sizeof(double) * NX); /* controller.py:416 */
        } else {
            for (int k = 0; k <= N; k++) memcpy(xt + (size_t)k * NX, x0i, sizeof(double) * NX);
            memset(ut, 0, sizeof(double) * (size_t)N * NU);
        }
        const double *yr = yref_bcast ? yref : yref + (size_t)ib * N * NY;
        const double *ye = yref_bcast ? yref_e : yref_e + (size_t)ib * NX;
        orc_stats st;
        const int s = orc_sqp_rti(c, x0i, yr, ye, xt, ut, &st);
        if (status) status[ib] = s;
        if (iters) iters[ib] = st.qp_iter;
        if (passes) passes[ib] = (s == 0 && st.polished) ? st.polish_attempts : -st.polish_attempts;   /* the library's convention (nmpc_device_passes) */
        if (growth) growth[ib] = st.growth;
        for (int i = 0; i < NU; i++) u0[(size_t)ib * NU + i] = (s == 0) ? ut[i] : 0.0; /* controller.py:448-450 */
        if (s != 0) { /* the warm start is invalidated (controller.py:448-450): hand back the cold-start point (:425-431) */
            for (int k = 0; k <= N; k++) memcpy(xt + (size_t)k * NX, x0i, sizeof(double) * NX);
            memset(ut, 0, sizeof(double) * (size_t)N * NU);
        }
        if (x_out) memcpy(x_out + (size_t)ib * (N + 1) * NX, xt, sizeof(double) * (size_t)(N + 1) * NX);
        if (u_out) memcpy(u_out + (size_t)ib * N * NU, ut, sizeof(double) * (size_t)N * NU);
        orc_free(xt);
        orc_free(ut);
        if (ORC_USE_ARENA && ar.need > ar.cap && ar.cap < ORC_ARENA_MAX) {             /* (some allocations of this instance fell back to the heap) */
            free(ar.base);
            ar.cap = ar.need + ar.need / 2;
            if (ar.cap > ORC_ARENA_MAX) ar.cap = ORC_ARENA_MAX;   /* (beyond that the heap serves the rest: long horizons, condensed blocks) */
            ar.base = (char *)malloc(ar.cap);
            if (!ar.base) ar.cap = 0;
        }
    }
        t_arena = NULL;
        free(ar.base);
    }
    return 0;
}
