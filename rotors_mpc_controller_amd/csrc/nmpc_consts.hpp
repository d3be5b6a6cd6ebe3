// nmpc_consts.hpp -- host-side derivation of the per-solver constant block from nmpc_config.
// (OCP definition controller.py:175-264; [UPSTREAM] U4 cost scaling, U5 Levenberg-Marquardt.)
#pragma once

#include <cstring>

#include "../../include/rotors_nmpc.h"
#include "nmpc_lane.hpp"

namespace nmpc {

// Attempt policy of the active-set passes where nmpc_config leaves it at 0 (the default): 8 passes per attempt, 16 in total below N = 160;
// from N = 160 up ONE attempt of 16-32 passes.  Why the long horizon differs (round 5, measured on config 5: N = 600, B = 1024, near hover):
// an interior-point iteration between two attempts is three sequential 600-stage solves (0.9 ms of a 7.3 ms solve) and re-derives the
// active set the passes were converging to anyway - with one attempt of 16 every instance of the sample is accepted without one (14 passes
// at most instead of 8 + 1 iteration + 5), and the commands are the same bits (an accepted pass is the exact solution of the same pinned
// problem).  On short horizons the split schedule stays: there the iteration is cheap and caps what a wave's slowest team costs its mates.
// How long that one attempt may be grows with the horizon: N / 16 passes, at least 16, at most 32 (N = 250: 16, N = 600: 32).  The pins of
// a long horizon are found by creeping (section 4.5 of DESIGN.md: the saturated stretch at the end of the horizon is entered from its far end,
// a few stages per pass), so the pass count scales with N: at N = 600 the near-hover samples need 13-14 passes at most on ten of twelve
// seeds but 20 and 23 on seeds 1 and 5 (ONE instance of 1 024 each), the aggressive sample 28; at N = 250 nothing needs more than 11.  An
// instance that runs out of passes leaves the block-parallel tail for the sequential interior point - 19-20 iterations over 600 stages,
// 25 ms where the whole batch takes 6: with 16 passes config 5 on the SURVEY's own sample (seed 5) took 31.3 ms, with 32 it takes 8.6
// (seed 1: 32.7 -> 7.9; aggressive: 34.8 -> 11.3); the sixteen extra steps cost a batch that does not need them 0.2 ms (seed 0: 5.84 -> 6.04;
// profiles/r05p_config5_budget.txt).
// The oracle applies the same rule (oracle/nmpc_oracle.c: orc_polish_policy).
inline int long_horizon_passes(int N) { const int p = N / 16; return p < 16 ? 16 : (p > 32 ? 32 : p); }
inline void resolve_polish_policy(int N, int &passes, int &budget)
{
    if (passes <= 0) passes = N >= 160 ? long_horizon_passes(N) : 8;
    if (budget <= 0) budget = N >= 160 ? (passes > 16 ? passes : 16) : 2 * passes;
}

template <class T>
void fill_consts(const nmpc_config &g, Consts<T> &c)
{
    std::memset(&c, 0, sizeof(c));
    c.N = g.N;
    c.steps = g.sim_num_steps;
    c.iter_max = g.qp_iter_max;
    c.shared = 0;
    c.dt = (T)g.dt;
    c.h = (T)(g.dt / g.sim_num_steps);
    const double sc = g.cost_scaled_by_dt ? g.dt : 1.0;                       // U4
    const double lmk = g.levenberg_marquardt * (g.lm_scaled_by_dt ? g.dt : 1.0);  // U5
    for (int i = 0; i < NX; i++) {
        c.Qd[i] = (T)(sc * g.W[i] + lmk);
        c.QdN[i] = (T)(g.W_e[i] + g.levenberg_marquardt);
        c.Wq[i] = (T)(sc * g.W[i]);
        c.WqN[i] = (T)g.W_e[i];
    }
    for (int i = 0; i < NU; i++) {
        c.Rd[i] = (T)(sc * g.W[NX + i] + lmk);
        c.Wr[i] = (T)(sc * g.W[NX + i]);
        c.lbu[i] = (T)g.lbu[i];
        c.ubu[i] = (T)g.ubu[i];
        c.rx[i] = (T)g.rotor_x[i];
        c.ry[i] = (T)g.rotor_y[i];
        c.rz[i] = (T)g.rotor_z[i];
        c.fuw[0][i] = (T)(g.rotor_y[i] / g.inertia[0]);
        c.fuw[1][i] = (T)(-g.rotor_x[i] / g.inertia[1]);
        c.fuw[2][i] = (T)(g.rotor_z[i] / g.inertia[2]);
    }
    c.inv_mass = (T)(1.0 / g.mass);
    c.gravity = (T)g.gravity;
    for (int i = 0; i < 3; i++) { c.J[i] = (T)g.inertia[i]; c.invJ[i] = (T)(1.0 / g.inertia[i]); }
    c.tol_comp = (T)g.qp_tol_comp;
    c.tol_stat = (T)g.qp_tol_stat;
    c.mu0 = (T)g.qp_mu0;
    c.tau = (T)g.qp_tau;
    c.thr0 = (T)g.qp_thr0;
    c.thr0_rel = (T)g.qp_thr0_rel;
    c.polish = g.qp_polish;
    c.polish_passes = g.qp_polish_passes;
    c.polish_budget = g.qp_polish_budget;
    resolve_polish_policy(g.N, c.polish_passes, c.polish_budget);
    c.polish_ckpt = g.qp_polish_ckpt < 0 ? 0 : (g.qp_polish_ckpt > g.N - 1 ? g.N - 1 : g.qp_polish_ckpt);
    c.polish_mu = (T)g.qp_polish_mu;
    c.kkt_tol = sizeof(T) == 8 ? (T)1e-9 : (T)1e-5;   // the oracle uses 1e-9; FP32 gradients carry ~1e-6 noise
    c.growth_max = (T)g.qp_growth_max;
    c.acc_comp = (T)g.qp_acc_comp;
    c.acc_stat = (T)g.qp_acc_stat;
    c.tol_step = (T)g.qp_tol_step;
    c.maxiter_status = g.qp_maxiter_status;
    c.warm_start = g.qp_warm_start;
}


}  // namespace nmpc
