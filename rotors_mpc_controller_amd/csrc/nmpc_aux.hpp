// nmpc_aux.hpp -- the steps either side of the solve, batched on the device so
// that a closed-loop Monte-Carlo run never leaves HBM.  All three are HBM-bound element-wise
// kernels; one thread per output element keeps every wave access contiguous.
//
//   k_hover_reference    ReferenceGenerator.build_horizon + the stacking of controller.py:433-445
//                        for constant setpoints: position [B][3], yaw [B] -> yref [B][N][17], yref_e [B][13]
//                        (reference.py:11-13,75-91)
//   k_odometry_to_state  MPCControllerNode._odom_cb (nodes/mpc_controller_node:88-113, :25-44, :138-150):
//                        body-frame twist -> world-frame velocity via the ZYX Euler round trip the
//                        node performs, quaternion reordered to (w,x,y,z); x0 [B][13]
//   k_motor_speeds       MPCControllerNode._publish_command (nodes/mpc_controller_node:152-164):
//                        clip to the input bounds, omega = sqrt(u / k_f), clip to the motor speed limits
#pragma once

#include <hip/hip_runtime.h>

#include "nmpc_lane.hpp"

namespace nmpc {

template <class T>
__global__ void k_hover_reference(int B, int N, const T *__restrict__ pos, const T *__restrict__ yaw, T thrust,
                                  T *__restrict__ yref, T *__restrict__ yref_e)
{
    const size_t n_stage = (size_t)B * N * 17, n_all = n_stage + (size_t)B * 13;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n_all; idx += (size_t)gridDim.x * blockDim.x) {
        int b, i;
        T *dst;
        if (idx < n_stage) { b = (int)(idx / ((size_t)N * 17)); i = (int)(idx % 17); dst = yref + idx; }
        else { const size_t e = idx - n_stage; b = (int)(e / 13); i = (int)(e % 13); dst = yref_e + e; }
        T v = 0;
        if (i < 3) v = pos[(size_t)b * 3 + i];
        else if (i == 6) v = cos(T(0.5) * yaw[b]);
        else if (i == 9) v = sin(T(0.5) * yaw[b]);
        else if (i >= 13) v = thrust;
        *dst = v;
    }
}

template <class T>
__global__ void k_odometry_to_state(int B, const T *__restrict__ pose /*[B][7] p(3), q(x,y,z,w)*/,
                                    const T *__restrict__ twist /*[B][6] body linear(3), angular(3)*/,
                                    T *__restrict__ x0 /*[B][13]*/)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const T *p = pose + (size_t)b * 7, *tw = twist + (size_t)b * 6;
    T qx = p[3], qy = p[4], qz = p[5], qw = p[6];
    T roll = 0, pitch = 0, yaw = 0;
    const T n = sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
    if (n != T(0)) {                                   // quaternion_to_euler, node:25-44
        const T ax = qx / n, ay = qy / n, az = qz / n, aw = qw / n;
        roll = atan2(T(2) * (aw * ax + ay * az), T(1) - T(2) * (ax * ax + ay * ay));
        const T sp = T(2) * (aw * ay - az * ax);
        pitch = fabs(sp) >= T(1) ? copysign(T(1.5707963267948966), sp) : asin(sp);
        yaw = atan2(T(2) * (aw * az + ax * ay), T(1) - T(2) * (ay * ay + az * az));
    }
    const T cr = cos(roll), sr = sin(roll), cp = cos(pitch), sp2 = sin(pitch), cy = cos(yaw), sy = sin(yaw);
    const T vx = tw[0], vy = tw[1], vz = tw[2];        // _rotation_matrix, node:138-150
    T *o = x0 + (size_t)b * 13;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
    o[3] = cp * cy * vx + (cy * sp2 * sr - sy * cr) * vy + (cy * sp2 * cr + sy * sr) * vz;
    o[4] = cp * sy * vx + (sy * sp2 * sr + cy * cr) * vy + (sy * sp2 * cr - cy * sr) * vz;
    o[5] = -sp2 * vx + cp * sr * vy + cp * cr * vz;
    o[6] = qw; o[7] = qx; o[8] = qy; o[9] = qz;        // (w,x,y,z), node:103-106; not normalised here
    o[10] = tw[3]; o[11] = tw[4]; o[12] = tw[5];
}

template <class T>
struct Bounds4 {
    T lb[4], ub[4];
};

template <class T>
__global__ void k_motor_speeds(int n /*B*4*/, const T *__restrict__ u, Bounds4<T> bd, T kf, T w_min, T w_max,
                               T *__restrict__ speeds, T *__restrict__ clipped /*nullable*/)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = i & 3;
    const T lbu = j == 0 ? bd.lb[0] : (j == 1 ? bd.lb[1] : (j == 2 ? bd.lb[2] : bd.lb[3]));
    const T ubu = j == 0 ? bd.ub[0] : (j == 1 ? bd.ub[1] : (j == 2 ? bd.ub[2] : bd.ub[3]));
    const T c = fmin(fmax(u[i], lbu), ubu);
    T s2 = c / fmax(kf, T(1e-9));
    s2 = fmin(fmax(s2, T(0)), w_max * w_max);
    speeds[i] = fmin(fmax(sqrt(s2), w_min), w_max);
    if (clipped) clipped[i] = c;
}

// What MPCControllerNode._loop does with a solve result (nodes/mpc_controller_node:122-131,152-164): a
// successful solve publishes its command clipped to the input bounds and remembers it as _last_command; a
// failed one (status != 0) re-publishes _last_command.  held [B][4] is that memory and, after the call,
// the command the plant receives.
template <class T>
__global__ void k_hold_command(int n /*B*4*/, const T *__restrict__ u0, const int32_t *__restrict__ status, Bounds4<T> bd,
                               T *__restrict__ held)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = i & 3;
    const T lbu = j == 0 ? bd.lb[0] : (j == 1 ? bd.lb[1] : (j == 2 ? bd.lb[2] : bd.lb[3]));
    const T ubu = j == 0 ? bd.ub[0] : (j == 1 ? bd.ub[1] : (j == 2 ? bd.ub[2] : bd.ub[3]));
    if (status[i >> 2] == 0) held[i] = fmin(fmax(u0[i], lbu), ubu);
}

// The plant of a closed-loop Monte-Carlo rollout: the same model and the same ERK scheme
// as the controller's prediction (controller.py:183-188, 267-355), one shooting interval, states only.
// normalize_q mirrors what PositionNMPC.solve does to every measured state (controller.py:406-409).
template <class T>
__global__ void k_plant_step(Consts<T> c, int B, const T *__restrict__ x, const T *__restrict__ u,
                             T *__restrict__ xn, int normalize_q)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    T xs[NX], us[NU], f1[NX], xm[NX], f2[NX];
    NMPC_UNROLL for (int i = 0; i < NX; i++) xs[i] = x[(size_t)b * NX + i];
    NMPC_UNROLL for (int i = 0; i < NU; i++) us[i] = u[(size_t)b * NU + i];
    for (int st = 0; st < c.steps; st++) {
        model_f(c, xs, us, f1);
        NMPC_UNROLL for (int i = 0; i < NX; i++) xm[i] = xs[i] + T(0.5) * c.h * f1[i];
        model_f(c, xm, us, f2);
        NMPC_UNROLL for (int i = 0; i < NX; i++) xs[i] += c.h * f2[i];
    }
    if (normalize_q) {
        const T n = sqrt(xs[6] * xs[6] + xs[7] * xs[7] + xs[8] * xs[8] + xs[9] * xs[9]);
        if (n != T(0)) { NMPC_UNROLL for (int i = 6; i < 10; i++) xs[i] /= n; }
    }
    NMPC_UNROLL for (int i = 0; i < NX; i++) xn[(size_t)b * NX + i] = xs[i];
}

// One closed-loop tick's plant side in ONE launch (rollout.py): hold the command (k_hold_command), fly it for one
// shooting interval (k_plant_step) and leave the next measured state in place of the old one.
template <class T>
__global__ void k_hold_and_step(Consts<T> c, int B, const T *__restrict__ u0, const int32_t *__restrict__ status,
                                Bounds4<T> bd, T *__restrict__ held, T *__restrict__ x, int normalize_q)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    T xs[NX], us[NU], f1[NX], xm[NX], f2[NX];
    NMPC_UNROLL for (int i = 0; i < NX; i++) xs[i] = x[(size_t)b * NX + i];
    const bool fresh = status[b] == 0;
    NMPC_UNROLL for (int j = 0; j < NU; j++) {
        const T h = held[(size_t)b * NU + j], v = fmin(fmax(u0[(size_t)b * NU + j], bd.lb[j]), bd.ub[j]);
        us[j] = fresh ? v : h;
        if (fresh) held[(size_t)b * NU + j] = v;
    }
    for (int st = 0; st < c.steps; st++) {
        model_f(c, xs, us, f1);
        NMPC_UNROLL for (int i = 0; i < NX; i++) xm[i] = xs[i] + T(0.5) * c.h * f1[i];
        model_f(c, xm, us, f2);
        NMPC_UNROLL for (int i = 0; i < NX; i++) xs[i] += c.h * f2[i];
    }
    if (normalize_q) {
        const T n = sqrt(xs[6] * xs[6] + xs[7] * xs[7] + xs[8] * xs[8] + xs[9] * xs[9]);
        if (n != T(0)) { NMPC_UNROLL for (int i = 6; i < 10; i++) xs[i] /= n; }
    }
    NMPC_UNROLL for (int i = 0; i < NX; i++) x[(size_t)b * NX + i] = xs[i];
}

// Adjoint sensitivities: model_adj / erk_adjoint live in nmpc_lane.hpp (host + device, so that the CPU build of the
// kernel bodies tests them too); the two kernels below batch them.
// out [B][17] = ( A' lam (13) | B' lam (4) ) of the interval that starts at (x, u); cont != 0: the continuous
// right-hand side ( f_x' lam | f_u' lam ) instead (what expl_vde_adj returns)
template <class T>
__global__ void k_adjoint_sens(Consts<T> c, int B, const T *__restrict__ x, const T *__restrict__ u, const T *__restrict__ lam,
                               T *__restrict__ out, int cont)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    T xs[NX], us[NU], l[NX], gu[NU];
    NMPC_UNROLL for (int i = 0; i < NX; i++) { xs[i] = x[(size_t)b * NX + i]; l[i] = lam[(size_t)b * NX + i]; }
    NMPC_UNROLL for (int i = 0; i < NU; i++) us[i] = u[(size_t)b * NU + i];
    if (cont) {
        Jac<T> J;
        T ax[NX];
        model_jac(c, xs, us, J);
        model_adj(c, J, l, ax, gu);
        NMPC_UNROLL for (int i = 0; i < NX; i++) l[i] = ax[i];
    } else {
        erk_adjoint<T, ADJ_MAX_STEPS>(c, xs, us, l, gu, nullptr);
    }
    NMPC_UNROLL for (int i = 0; i < NX; i++) out[(size_t)b * (NX + NU) + i] = l[i];
    NMPC_UNROLL for (int i = 0; i < NU; i++) out[(size_t)b * (NX + NU) + NX + i] = gu[i];
}

// Stationarity / feasibility report of a trajectory (x [B][N+1][13], u [B][N][4]) for the NLP of controller.py:175-264,
// by ONE adjoint sweep per instance (no A, B):  lam_N = W_e (x_N - yref_e);  k = N-1..0:  g_u = W_u (u_k - yref_u) + B_k' lam_{k+1},
// lam_k = W_x (x_k - yref_x) + A_k' lam_{k+1}.  res [B][3] = ( max_k |projected g_u| with the input box, max_k |phi(x_k,u_k) - x_{k+1}|,
// max_k bound violation ).  The projection: an input on its lower (upper) bound contributes only the negative (positive) part
// of g_u.  It is what tells a converged SQP iterate from an RTI step that still has a way to go.
template <class T>
__global__ void k_kkt_report(Consts<T> c, int B, const T *__restrict__ x, const T *__restrict__ u, const T *__restrict__ yref,
                             const T *__restrict__ yref_e, int yref_bcast, T *__restrict__ res)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int N = c.N;
    const T *xb = x + (size_t)b * (N + 1) * NX, *ub = u + (size_t)b * N * NU;
    const T *yr = yref_bcast ? yref : yref + (size_t)b * N * NY, *ye = yref_bcast ? yref_e : yref_e + (size_t)b * NX;
    T lam[NX];
    NMPC_UNROLL for (int i = 0; i < NX; i++) lam[i] = c.WqN[i] * (xb[(size_t)N * NX + i] - ye[i]);
    T rs = 0, rd = 0, rb = 0;
    for (int k = N - 1; k >= 0; k--) {
        T xs[NX], us[NU], gu[NU], xn[NX];
        NMPC_UNROLL for (int i = 0; i < NX; i++) xs[i] = xb[(size_t)k * NX + i];
        NMPC_UNROLL for (int i = 0; i < NU; i++) us[i] = ub[(size_t)k * NU + i];
        erk_adjoint<T, ADJ_MAX_STEPS>(c, xs, us, lam, gu, xn);
        NMPC_UNROLL for (int i = 0; i < NX; i++) {
            lam[i] += c.Wq[i] * (xs[i] - yr[(size_t)k * NY + i]);
            rd = fmax(rd, fabs(xn[i] - xb[(size_t)(k + 1) * NX + i]));
        }
        NMPC_UNROLL for (int j = 0; j < NU; j++) {
            const T g = gu[j] + c.Wr[j] * (us[j] - yr[(size_t)k * NY + NX + j]);
            const T tol = T(1e-9) * (T(1) + fabs(c.lbu[j]) + fabs(c.ubu[j]));
            T pg = g;
            if (us[j] <= c.lbu[j] + tol) pg = fmin(g, T(0));           // at the lower bound only a negative gradient is a violation
            else if (us[j] >= c.ubu[j] - tol) pg = fmax(g, T(0));
            rs = fmax(rs, fabs(pg));
            rb = fmax(rb, fmax(c.lbu[j] - us[j], us[j] - c.ubu[j]));
        }
    }
    res[(size_t)b * 3] = rs; res[(size_t)b * 3 + 1] = rd; res[(size_t)b * 3 + 2] = fmax(rb, T(0));
}

}  // namespace nmpc
