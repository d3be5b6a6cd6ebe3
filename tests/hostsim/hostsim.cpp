// hostsim.cpp -- TEST-ONLY harness: compiles the per-lane kernel bodies of
// rotors_mpc_controller_amd/csrc/nmpc_{lane,ipm}.hpp for the host and runs them lane by lane
// over the same SoA workspace layout the GPU uses.  Lets the `-m "not gpu"` suite check the
// kernel arithmetic against the oracle without a GPU.  It is NOT linked into, loaded by, or
// reachable from the product library (librotors_nmpc_hip.so has no CPU path).
#include <cstdint>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <cstring>
#include <vector>

#include "../../include/rotors_nmpc.h"
#include "../../rotors_mpc_controller_amd/csrc/nmpc_consts.hpp"
#include "../../rotors_mpc_controller_amd/csrc/nmpc_cond.hpp"
#include "../../rotors_mpc_controller_amd/csrc/nmpc_ipm.hpp"

using namespace nmpc;

template <class T>
static void run(const nmpc_config &g, int B, const double *x0, const double *yref, const double *yref_e,
                int bcast, const double *x_init, const double *u_init, double *u0, int32_t *status,
                double *x_out, double *u_out, int32_t *iters, int shared)
{
    // A thread owns CH lanes at a time in a PRIVATE workspace of CH lanes (row stride CH: one cache line per row, the whole
    // workspace ~0.4 MB at N = 20, resident in its L2) - round 3 ran all lanes in one [rows][Bp] workspace, where a lane's rows
    // lie Bp * 8 bytes apart (a TLB entry per row at Bp = 16384) and scaled 3.6x on 256 cores.  Nothing is allocated inside the
    // loop over chunks.  Arithmetic per lane is unchanged: lane kernels only use w.Bp as the row stride.
    constexpr int CH = 8;
    const size_t N = g.N, Bp = CH;
    Consts<T> c;
    fill_consts(g, c);
    c.shared = (shared && !x_init) ? 1 : 0;
    auto cv = [](const double *p, size_t n) { std::vector<T> v(n); for (size_t i = 0; i < n; i++) v[i] = (T)p[i]; return v; };
    std::vector<T> hx0 = cv(x0, (size_t)B * NX), hy = cv(yref, (bcast ? 1 : B) * N * NY),
                   hye = cv(yref_e, (bcast ? 1 : B) * NX);
    std::vector<T> hxi, hui;
    if (x_init) { hxi = cv(x_init, (size_t)B * (N + 1) * NX); hui = cv(u_init, (size_t)B * N * NU); }
    std::vector<int32_t> it(B), st(B);
    std::vector<T> ou0((size_t)B * NU), oxo((size_t)B * (N + 1) * NX), ouo((size_t)B * N * NU);
    const bool cond = (g.flags & NMPC_FLAG_CONDENSED_QP) != 0;
    CondWork<T> cw0;
    const int N2 = (g.qp_cond_N > 0 && g.qp_cond_N < g.N) ? g.qp_cond_N : g.N;
    const size_t cond_n = cond ? (size_t)cond_layout(cw0, g.N, N2) * Bp : 0;
    // (instances are independent; OpenMP over chunks of lanes is what bench.py's second cpu_baseline row times)
#pragma omp parallel
    {
        std::vector<T> AB(N * AB_ROWS * Bp), bv(N * NX * Bp), qr((N * QR_ROWS + NX) * Bp), xl((N + 1) * NX * Bp),
            ul(N * NU * Bp), LM(N * LM_ROWS * Bp), iv(N * IV_ROWS * Bp), cbuf(cond_n);
        int32_t itc[CH], stc[CH];
        CondWork<T> cw = cw0;
        cw.base = cbuf.data();
        cw.Bp = (int)Bp;
#pragma omp for schedule(static)
        for (int c0 = 0; c0 < B; c0 += CH) {
            const int n = B - c0 < CH ? B - c0 : CH;
            Work<T> w{(int)Bp, AB.data(), bv.data(), qr.data(), xl.data(), ul.data(), LM.data(), iv.data(), itc, stc, nullptr, nullptr, nullptr};
            Inputs<T> in{hx0.data() + (size_t)c0 * NX, bcast ? hy.data() : hy.data() + (size_t)c0 * N * NY,
                         bcast ? hye.data() : hye.data() + (size_t)c0 * NX,
                         x_init ? hxi.data() + (size_t)c0 * (N + 1) * NX : nullptr, x_init ? hui.data() + (size_t)c0 * N * NU : nullptr, bcast};
            Outputs<T> out{ou0.data() + (size_t)c0 * NU, oxo.data() + (size_t)c0 * (N + 1) * NX, ouo.data() + (size_t)c0 * N * NU};
            for (int lane = 0; lane < n; lane++) {
                lane_prepare(c, w, in, lane);
                if (cond) lane_cond_ipm(c, w, cw, in, out, lane); else lane_ipm(c, w, in, out, lane);
                it[c0 + lane] = itc[lane]; st[c0 + lane] = stc[lane];
            }
        }
    }
    for (size_t i = 0; i < ou0.size(); i++) u0[i] = ou0[i];
    if (x_out) for (size_t i = 0; i < oxo.size(); i++) x_out[i] = oxo[i];
    if (u_out) for (size_t i = 0; i < ouo.size(); i++) u_out[i] = ouo[i];
    for (int i = 0; i < B; i++) { if (status) status[i] = st[i]; if (iters) iters[i] = it[i]; }
}

extern "C" void hostsim_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

extern "C" int hostsim_solve_batch(const nmpc_config *g, int B, const double *x0, const double *yref,
                                   const double *yref_e, int bcast, const double *x_init, const double *u_init,
                                   double *u0, int32_t *status, double *x_out, double *u_out, int32_t *iters)
{
    const int shared = (g->flags & NMPC_FLAG_SHARE_COLD_START) ? 1 : 0;
    run<double>(*g, B, x0, yref, yref_e, bcast, x_init, u_init, u0, status, x_out, u_out, iters, shared);     // the arithmetic is FP64 for every dtype
    return 0;
}

// adjoint sensitivities (nmpc_lane.hpp model_adj / erk_adjoint) of one interval per instance, FP64:
// out [B][17] = (A' lam | B' lam), or (f_x' lam | f_u' lam) when cont != 0
extern "C" int hostsim_adjoint(const nmpc_config *g, int B, const double *x, const double *u, const double *lam, double *out, int cont)
{
    Consts<double> c;
    fill_consts(*g, c);
    for (int b = 0; b < B; b++) {
        double l[NX], gu[NU];
        for (int i = 0; i < NX; i++) l[i] = lam[(size_t)b * NX + i];
        if (cont) {
            Jac<double> J;
            double ax[NX];
            model_jac(c, x + (size_t)b * NX, u + (size_t)b * NU, J);
            model_adj(c, J, l, ax, gu);
            for (int i = 0; i < NX; i++) l[i] = ax[i];
        } else {
            erk_adjoint<double, ADJ_MAX_STEPS>(c, x + (size_t)b * NX, u + (size_t)b * NU, l, gu, nullptr);
        }
        for (int i = 0; i < NX; i++) out[(size_t)b * (NX + NU) + i] = l[i];
        for (int i = 0; i < NU; i++) out[(size_t)b * (NX + NU) + NX + i] = gu[i];
    }
    return 0;
}
