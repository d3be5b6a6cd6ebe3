// nmpc_capi.hip -- HIP kernels (gfx950) and the C ABI of include/rotors_nmpc.h.
//
// Kernels defined here: k_prepare / k_ipm (one instance per lane: the mapping BASELINE.json's north_star names, kept as the
// fidelity path of the lane layout), k_cond_ipm (partial condensing as acados configures HPIPM, fidelity path) and the
// element-wise kernels of nmpc_aux.hpp.  The kernels that run by default - k_team_as, k_team_qp, k_team_qp_list, k_team_tail
// (16 lanes per MPC instance, 4 instances per wave, FP64 Riccati sweeps on v_mfma_f64_4x4x4 register tiles) and the block
// sweeps - live in nmpc_as.hip / nmpc_qp(f).hip / nmpc_block(f).hip.  Every workgroup is one 64-lane wave, so a batch becomes
// hundreds to thousands of independent workgroups that the dispatcher spreads over all XCDs; no
// inter-workgroup communication exists on this path (instances are independent), so no
// release/acquire protocol is needed.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/rotors_nmpc.h"
#include "nmpc_ipm.hpp"
#include "nmpc_team.hpp"
#include "nmpc_team_as.hpp"
#include "nmpc_as_launch.hpp"
#include "nmpc_block_launch.hpp"
#include "nmpc_cond.hpp"
#include "nmpc_aux.hpp"
#include "nmpc_consts.hpp"

using namespace nmpc;

namespace {

thread_local std::string g_create_error;

template <class T>
__global__ __launch_bounds__(64) void k_prepare(Consts<T> c, Work<T> w, Inputs<T> in, int B)
{
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane < B) lane_prepare(c, w, in, lane);
}

template <class T>
__global__ __launch_bounds__(64) void k_ipm(Consts<T> c, Work<T> w, Inputs<T> in, Outputs<T> out, int B)
{
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane < B) lane_ipm(c, w, in, out, lane);
}

// QP phase as acados configures it: partial condensing + IPM on the condensed QP (one instance per lane)
template <class T>
__global__ __launch_bounds__(64) void k_cond_ipm(Consts<T> c, Work<T> w, CondWork<T> cw, Inputs<T> in, Outputs<T> out, int B)
{
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane < B) lane_cond_ipm(c, w, cw, in, out, lane);
}

}  // namespace

struct nmpc_solver {
    nmpc_config cfg;
    std::string err;
    int Bp = 0;
    size_t esz = 8;   // element size of the caller-facing device buffers
    size_t wsz = 8;   // element size of the workspace = of the arithmetic
    // device workspace (element type = cfg.dtype)
    void *AB = nullptr, *bv = nullptr, *qr = nullptr, *xl = nullptr, *ul = nullptr, *LM = nullptr, *iv = nullptr, *tAB = nullptr, *tP = nullptr, *cond = nullptr;
    int32_t *d_iters = nullptr, *d_status = nullptr, *d_npol = nullptr;
    int *d_wl = nullptr;             // work list of the split FP64 path: count | done | list [Bp]
    double *d_gbase = nullptr;       // growth certificate: first-factorisation value per instance (active-set kernel -> work-list launch)
    void *d_consts = nullptr;        // Consts<double> in device memory (the active-set kernel reads it from there)
    int team_split = 1;              // active-set kernel + work-list launch (default); NMPC_TEAM_SPLIT=0: one general kernel
    int list_grid = 256;             // workgroups of the work-list launch (NMPC_LIST_GRID; each strides over the list.  64 until round 4: a list of
                                     // 1500 instances then took 1.97 ms against 0.93 with 256, and an empty list costs the same either way)
    int team_inplace = 1;            // a failed first attempt continues on its own wave inside k_team_as (no work-list launch); NMPC_TEAM_INPLACE=0:
                                     // work list + k_team_qp_list (always so for long horizons - the block-parallel tail - and the 256-register builds)
    int as_noflag = 0;               // NMPC_AS_NOFLAG=1: k_team_as from nmpc_qp.hip (default code generation) instead of nmpc_as.hip
    int block_noflag = 0;            // NMPC_BLOCK_NOFLAG=1: block kernels from nmpc_block.hip (default code generation) instead of nmpc_blockf.hip
    int lds_pad = 24;                // NMPC_LDS_PAD=<0..31> (experiments): team stride mod 32 doubles
    int lds_overlap = 1;             // NMPC_LDS_OVERLAP=0: the per-stage variant's stage cache behind the evaluation-point buffer (round 3)
    int qp_noflag = 0;               // NMPC_QP_NOFLAG=1: k_team_qp from nmpc_qp.hip (default code generation) instead of nmpc_qpf.hip
    int as_v256 = 0;                 // NMPC_AS_BUILD=v256: the 256-register build with the LDS stage cache (nmpc_qp.hip, OCC = 3)
    int team_lstg = -1;              // NMPC_TEAM_LSTG caps the stages whose factors stay in LDS (experiments; -1 = what fits)
    long long *d_prof = nullptr;   // only allocated in NMPC_PROFILE builds
    // device staging for the host-pointer entry points
    void *s_x0 = nullptr, *s_yref = nullptr, *s_yref_e = nullptr, *s_xi = nullptr, *s_ui = nullptr;
    void *s_u0 = nullptr, *s_xo = nullptr, *s_uo = nullptr;
    size_t staged_batch = 0;
    // small batches (and the single-instance nmpc_solve): ONE pinned host block in, one out, one device block
    // each, two asynchronous copies on the solver's own stream instead of up to nine pageable ones
    static constexpr int PACK_B = 64;
    void *h_in = nullptr, *h_out = nullptr, *d_in = nullptr, *d_out = nullptr;
    size_t pack_in_bytes = 0, pack_out_bytes = 0;
    hipStream_t pack_stream = nullptr;
    bool pack_ready = false;
    uint64_t ws_bytes = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // start | after prepare | after the (first) solve kernel | after the work-list launch
    bool timed_split = false, last_split = false, last_inplace = false, last_tail = false;
    bool last_shared = false;        // the last solve linearised ONE interval (shared cold start): tAB then holds stage-0 tiles only
    int last_B = 0;
    bool timed = false, timed_fused = false, solved = false;
    bool timing = true;   // nmpc_set_timing: HIP events around the kernels of every solve
    // single-instance slot (AcadosOcpSolver.set/get state)
    std::vector<double> sx, su, syref, syref_e, sx0;
    std::vector<float> cvt;   // FP32 conversions of the host-buffer entry point
    std::vector<double> sxo, suo;   // result slot of nmpc_solve (swapped in on success)
    // parallel-in-time factorisation (nmpc_block.hpp): buffers grown on demand
    double *blk_agg = nullptr, *blk_bnd = nullptr, *blk_chk = nullptr, *blk_fac = nullptr;
    int blk_J = 0;
    hipEvent_t blk_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // block-parallel tail of long-horizon solves: NMPC_BLOCK_TAIL = 0 off | 1 on | unset: on from N = 160 up (measured at B = 1024: N = 120 0.24 -> 0.46 ms,
    // 150 0.72 -> 0.70, 180 1.46 -> 1.18, 250 3.33 -> 2.29, 600 17.2 -> 10.1);
    // NMPC_BLOCK_J = blocks (unset: ~0.7 sqrt(N): measured optimum of config 5, 10.0 ms at J = 16-17 against 10.3 at 21 and 10.6 at 10)
    int block_tail = -1, block_J = 0, tail_cap = 1;   // NMPC_TAIL_CAP: passes of the first attempt the first launch performs itself
    int tail_J = 0, tail_M = 0;      // blocks that hold stages, stages per block (0: the tail is not used by this handle)
    double *d_ts = nullptr, *d_binfo = nullptr, *tail_agg = nullptr, *tail_bnd = nullptr;
    double *tail_gbuf = nullptr, *tail_xb = nullptr, *tail_frec = nullptr;   // block-parallel forward sweep (NMPC_TAIL_FWD=0: sequential)
    int tail_fwd = 1, tail_keep = 1;          // NMPC_TAIL_KEEP=0: every pass re-aggregates every block
    int tail_fwd_overlap = 1;                 // NMPC_TAIL_FWD_OVERLAP=0: the scan's forward walk at the end of the scan kernel (round 4) instead of beside the final sweeps
    int *d_wl2 = nullptr;            // fallback list of the tail: count | done | list [Bp]
    int *d_wl3 = nullptr;            // second work list of the tail (the list is compacted from step to step, alternating with d_wl)
    int team_occ = 0;   // 0 = default; NMPC_TEAM_OCC=1|2 picks the register budget variant
    int team_tpw = 0;   // 0 = choose from the batch size; NMPC_TEAM_TPW=1|2|4 overrides (experiments)

    // canaries around every device allocation of the handle (NMPC_GUARD=<KiB>, off by default): nmpc_debug_guard_check
    size_t guard = 0;
    struct GuardRec { char *base; size_t n; };
    std::vector<GuardRec> guards;

    int fail(int code, const char *fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
};


// Device allocations of a handle.  With NMPC_GUARD=<KiB> in the environment at create, every buffer sits between two canary
// bands of that size filled with 0xA5; nmpc_debug_guard_check() reports any byte of a band that a kernel (or a copy) has changed -
// a store outside a buffer that stays inside mapped memory does not fault, it corrupts a neighbour, and this is how it shows.
static hipError_t gmalloc(nmpc_solver *s, void **p, size_t n)
{
    if (!s->guard) return hipMalloc(p, n);
    char *b = nullptr;
    hipError_t e = hipMalloc((void **)&b, n + 2 * s->guard);
    if (e != hipSuccess) return e;
    e = hipMemset(b, 0xA5, s->guard);
    if (e == hipSuccess) e = hipMemset(b + s->guard + n, 0xA5, s->guard);
    if (e != hipSuccess) { (void)hipFree(b); return e; }
    s->guards.push_back({b, n});
    *p = b + s->guard;
    return hipSuccess;
}
static hipError_t gfree(nmpc_solver *s, void *p)
{
    if (s->guard) {
        for (size_t i = 0; i < s->guards.size(); i++)
            if (s->guards[i].base + s->guard == (char *)p) {
                char *b = s->guards[i].base;
                s->guards.erase(s->guards.begin() + i);
                return hipFree(b);
            }
    }
    return hipFree(p);
}

#define HIP_TRY(s, call)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) return (s)->fail(NMPC_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

extern "C" {

#ifndef NMPC_SOURCE_HASH
#define NMPC_SOURCE_HASH "unknown"
#endif
// NMPC_DEFAULT_NOFLAG (csrc/Makefile, tools/emu/codegen_gate.py): 1 when the build-time scan of the flag builds' assembly found something or hipcc
// is not the validated version - the default-codegen twins of every kernel then run by default (the NMPC_*_NOFLAG switches of nmpc_create)
#ifndef NMPC_DEFAULT_NOFLAG
#define NMPC_DEFAULT_NOFLAG 0
#endif
#if NMPC_DEFAULT_NOFLAG
#define NMPC_CODEGEN_NAME "default"
#else
#define NMPC_CODEGEN_NAME "flag"
#endif
// "rotors_nmpc_hip <abi> (gfx950) src <hash of the kernel sources this binary was built from> codegen <flag|default>": which builds of the
// kernels run by default - "flag": the ones compiled with the two internal LLVM options, which passed the build's gate; "default": plain -O3
const char *nmpc_version(void) { return "rotors_nmpc_hip 0.3 (gfx950) src " NMPC_SOURCE_HASH " codegen " NMPC_CODEGEN_NAME; }
// sizes of the two public structs in THIS binary: a binding checks them against its own mirror before it trusts either
// (nmpc_get_stats writes sizeof(nmpc_stats) bytes into the caller's buffer)
int nmpc_abi_sizes(int *config_bytes, int *stats_bytes)
{
    if (config_bytes) *config_bytes = (int)sizeof(nmpc_config);
    if (stats_bytes) *stats_bytes = (int)sizeof(nmpc_stats);
    return 3;            // ABI revision: 3 = round 3 (qp_maxiter_status, certificate fields in nmpc_config; n_tail / ms_tail in nmpc_stats)
}

void nmpc_default_config(nmpc_config *c)
{
    // reference config/params.yaml:1-33 and controller.py:98-110
    static const double W[NMPC_NY] = {10, 10, 8, 1, 1, 0.2, 3.2, 3.2, 3.2, 3.2, 1.4, 1.4, 0.4, 1.75, 1.75, 1.75, 1.75};
    static const double We[NMPC_NX] = {5, 5, 3, 2, 2, 2, 12, 12, 12, 18.5, 2, 2, 1.8};
    const double L = 0.17, kf = 8.54858e-6, km = 0.016;
    const double spin[4] = {-1, 1, -1, 1}, rx[4] = {L, 0, -L, 0}, ry[4] = {0, L, 0, -L};
    std::memset(c, 0, sizeof(*c));
    c->N = 20;
    c->dt = 0.05;
    std::memcpy(c->W, W, sizeof(W));
    std::memcpy(c->W_e, We, sizeof(We));
    for (int i = 0; i < 4; i++) {
        c->lbu[i] = kf * 50.0 * 50.0;
        c->ubu[i] = kf * 838.0 * 838.0;
        c->rotor_x[i] = rx[i];
        c->rotor_y[i] = ry[i];
        c->rotor_z[i] = spin[i] * km;
    }
    c->levenberg_marquardt = 7.0e-3;
    c->lm_scaled_by_dt = 1;
    c->cost_scaled_by_dt = 1;
    c->mass = 0.68;
    c->gravity = 9.81;
    c->inertia[0] = 0.007; c->inertia[1] = 0.007; c->inertia[2] = 0.012;
    c->sim_num_stages = 2;
    c->sim_num_steps = 2;
    c->qp_iter_max = 600;
    c->qp_cond_N = 5;
    c->qp_tol_comp = 1e-11;
    c->qp_tol_stat = 1e-11;
    c->qp_mu0 = 0.1;
    c->qp_tau = 0.995;
    c->qp_thr0 = 0.1;
    c->qp_thr0_rel = 0.25;
    c->dtype = NMPC_DTYPE_F64;
    c->device = 0;
    c->max_batch = 4096;
    c->flags = NMPC_FLAG_SHARE_COLD_START | NMPC_FLAG_TEAM_MAPPING;
    c->qp_polish = 1;
    c->qp_polish_passes = 0;   // 0 = the measured policy for the horizon (resolve_polish_policy, nmpc_consts.hpp): 8 per attempt / 16 in total
    c->qp_polish_budget = 0;   // below N = 160, one attempt of N / 16 passes (16 .. 32) from there up
    c->qp_polish_mu = 1.0;
    c->qp_polish_ckpt = 12;
    c->qp_maxiter_status = 0;
    c->qp_growth_max = 1e6;
    c->qp_acc_comp = 1e-8;
    c->qp_acc_stat = 1e-8;
    c->qp_tol_step = 1e-3;
    c->qp_warm_start = 1;
    c->reserved_ = 0;
}

static int ckpt_stages(const nmpc_config &g)
{
    return g.qp_polish_ckpt < 0 ? 0 : (g.qp_polish_ckpt > g.N - 1 ? g.N - 1 : g.qp_polish_ckpt);
}

static int alloc_ws(nmpc_solver *s)
{
    const size_t N = (size_t)s->cfg.N, Bp = (size_t)s->Bp, e = s->wsz;
    const size_t Bw = Bp + 1;      // per-instance team workspaces carry one spare row: idle teams of a wave work there
    // lane layout ([row][Bp]: k_prepare / k_ipm / k_cond_ipm) and team layout ([inst][rows]) need different arrays; a handle runs one of them
    const bool team = (s->cfg.flags & NMPC_FLAG_TEAM_MAPPING) && !(s->cfg.flags & NMPC_FLAG_CONDENSED_QP);
    const size_t ln = team ? 0 : 1;
    struct { void **p; size_t n; } a[] = {
        {&s->AB, std::max<size_t>(1, ln * N * AB_ROWS * Bp) * e}, {&s->bv, std::max<size_t>(1, ln * N * NX * Bp) * e},
        {&s->qr, std::max<size_t>(1, ln * (N * QR_ROWS + NX) * Bp) * e},
        {&s->xl, std::max<size_t>(1, ln * (N + 1) * NX * Bp) * e}, {&s->ul, std::max<size_t>(1, ln * N * NU * Bp) * e},
        {&s->LM, N * (team ? (size_t)TLM_ROWS : (size_t)LM_ROWS) * Bw * e},
        {&s->iv, N * IV_ROWS * Bw * e},
        {&s->tAB, (team ? N * TAB_ROWS * Bw : 1) * e},
        {&s->tP, (team && s->cfg.qp_polish ? (size_t)(ckpt_stages(s->cfg) + 1) * TP_ROWS * Bw : 1) * e},
        {(void **)&s->d_iters, Bp * sizeof(int32_t)},
        {(void **)&s->d_status, Bp * sizeof(int32_t)}, {(void **)&s->d_npol, Bp * sizeof(int32_t)},
        {(void **)&s->d_wl, (Bp + 2) * sizeof(int)}, {(void **)&s->d_gbase, (Bp + 1) * sizeof(double)}};
    for (auto &x : a) {
        HIP_TRY(s, gmalloc(s, x.p, x.n));
        s->ws_bytes += x.n;
    }
    HIP_TRY(s, hipMemset(s->d_wl, 0, (Bp + 2) * sizeof(int)));
    {
        Consts<double> cd;
        fill_consts(s->cfg, cd);
        HIP_TRY(s, gmalloc(s, &s->d_consts, sizeof(cd)));
        HIP_TRY(s, hipMemcpy(s->d_consts, &cd, sizeof(cd), hipMemcpyHostToDevice));
    }
    if (s->cfg.flags & NMPC_FLAG_CONDENSED_QP) {
        CondWork<double> cw;
        const int N2 = (s->cfg.qp_cond_N > 0 && s->cfg.qp_cond_N < s->cfg.N) ? s->cfg.qp_cond_N : s->cfg.N;
        const size_t n = (size_t)cond_layout(cw, s->cfg.N, N2) * Bp * e;
        HIP_TRY(s, gmalloc(s, &s->cond, n));
        s->ws_bytes += n;
    }
    return 0;
}

nmpc_solver *nmpc_create(const nmpc_config *cfg)
{
    auto bad = [](const char *m) { g_create_error = m; return (nmpc_solver *)nullptr; };
    if (!cfg) return bad("nmpc_create: cfg is NULL");
    if (cfg->N < 1 || cfg->N > 4096) return bad("nmpc_create: N out of range [1,4096]");
    if (!(cfg->dt > 0)) return bad("nmpc_create: dt must be positive");
    if (cfg->sim_num_stages != 2)
        return bad("nmpc_create: only sim_method_num_stages = 2 (explicit midpoint, controller.py:187) is built");
    if (cfg->sim_num_steps < 1) return bad("nmpc_create: sim_num_steps must be >= 1");
    if (cfg->dtype != NMPC_DTYPE_F64 && cfg->dtype != NMPC_DTYPE_F32 && cfg->dtype != NMPC_DTYPE_F32IO) return bad("nmpc_create: bad dtype");
    // FP32 buffers (NMPC_DTYPE_F32 = NMPC_DTYPE_F32IO since round 5: the arithmetic is FP64 either way, include/rotors_nmpc.h): every kernel of the
    // team mapping is instantiated for float arrays; the fidelity kernels of the lane layout are not
    if (cfg->dtype != NMPC_DTYPE_F64) {
        const bool ok = (cfg->flags & NMPC_FLAG_TEAM_MAPPING) && !(cfg->flags & NMPC_FLAG_CONDENSED_QP);
        if (!ok) return bad("nmpc_create: FP32 buffers (NMPC_DTYPE_F32 / NMPC_DTYPE_F32IO) need the team mapping without condensing (the default)");
    }
    if ((cfg->flags & NMPC_FLAG_TEAM_MAPPING) && !(cfg->flags & NMPC_FLAG_CONDENSED_QP) && cfg->sim_num_steps > AS_MAX_STEPS)
        return bad("nmpc_create: the team mapping is built for sim_num_steps <= 4 (controller.py:188 sets 2); clear NMPC_FLAG_TEAM_MAPPING for more");
    if (cfg->max_batch < 1) return bad("nmpc_create: max_batch must be >= 1");
    if (!(cfg->mass > 0) || !(cfg->inertia[0] > 0) || !(cfg->inertia[1] > 0) || !(cfg->inertia[2] > 0))
        return bad("nmpc_create: mass and inertia must be positive");
    for (int i = 0; i < NMPC_NU; i++)
        if (!(cfg->ubu[i] > cfg->lbu[i])) return bad("nmpc_create: need lbu < ubu");
    // PROJECT_REDUC_HESS (controller.py:189) as a check, not a transformation (U6)
    const double sc = cfg->cost_scaled_by_dt ? cfg->dt : 1.0;
    const double lmk = cfg->levenberg_marquardt * (cfg->lm_scaled_by_dt ? cfg->dt : 1.0);
    for (int i = 0; i < NMPC_NY; i++)
        if (!(sc * cfg->W[i] + lmk > 0)) return bad("nmpc_create: stage Hessian not positive definite (would need projection)");
    for (int i = 0; i < NMPC_NX; i++)
        if (!(cfg->W_e[i] + cfg->levenberg_marquardt > 0)) return bad("nmpc_create: terminal Hessian not positive definite");
    int ndev = 0;
    const hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev < 1) {
        g_create_error = std::string("nmpc_create: no HIP device visible (hipGetDeviceCount: ") +
                         hipGetErrorString(de) + ", count " + std::to_string(ndev) +
                         "); this library has no CPU path";
        return nullptr;
    }
    if (cfg->device < 0 || cfg->device >= ndev) return bad("nmpc_create: device ordinal out of range");
    if (hipSetDevice(cfg->device) != hipSuccess) return bad("nmpc_create: hipSetDevice failed");
    auto *s = new nmpc_solver();
    s->cfg = *cfg;
    // Attempt policy of the active-set passes when left at 0: 8 passes per attempt, 16 in total.  An extra pass costs the
    // instance ~10-25 us in the active-set kernel, a hand-over to the interior-point iteration ~0.5 ms for the whole batch
    // (measured on MI355X, B = 4096: aggressive set 6.6 M solves/s with 5 / 8 - 31 instances in the second launch - against
    // 18.7 M with 8 / 16, none; N = 250 near-hover 14.9 ms against 3.5 ms; N = 600 43.3 ms against 39.6 ms; more than 8 / 16
    // changed nothing or lost: N = 600 12 / 24 42.8 ms).  The near-hover N = 20 set needs 3 passes at most either way.
    // ... from N = 160 up: ONE attempt of 16 passes (nmpc_consts.hpp: resolve_polish_policy has the measurement)
    resolve_polish_policy(s->cfg.N, s->cfg.qp_polish_passes, s->cfg.qp_polish_budget);
    if (cfg->dtype == NMPC_DTYPE_F32) s->cfg.dtype = NMPC_DTYPE_F32IO;     // one FP32-buffer path
    s->esz = cfg->dtype == NMPC_DTYPE_F64 ? 8 : 4;
    s->wsz = 8;                                                             // workspace and arithmetic are always double
    if (const char *e = std::getenv("NMPC_TEAM_OCC")) {
        const int v = std::atoi(e);
        if (v == 1 || v == 2) s->team_occ = v;
    }
    if (const char *e = std::getenv("NMPC_GUARD")) s->guard = (size_t)std::max(0, std::atoi(e)) * 1024;
    if (const char *e = std::getenv("NMPC_TEAM_SPLIT")) s->team_split = std::atoi(e) != 0;
    if (const char *e = std::getenv("NMPC_TEAM_INPLACE")) s->team_inplace = std::atoi(e) != 0;
    if (const char *e = std::getenv("NMPC_LIST_GRID")) s->list_grid = std::max(1, std::min(4096, std::atoi(e)));
    s->as_noflag = s->qp_noflag = s->block_noflag = NMPC_DEFAULT_NOFLAG;        // the build's codegen gate; the three switches below override it either way
    if (const char *e = std::getenv("NMPC_AS_NOFLAG")) s->as_noflag = std::atoi(e) != 0;
    if (const char *e = std::getenv("NMPC_QP_NOFLAG")) s->qp_noflag = std::atoi(e) != 0;
    if (const char *e = std::getenv("NMPC_LDS_OVERLAP")) s->lds_overlap = std::atoi(e) != 0;
    if (const char *e = std::getenv("NMPC_LDS_PAD")) s->lds_pad = std::max(0, std::min(31, std::atoi(e)));
    if (const char *e = std::getenv("NMPC_BLOCK_NOFLAG")) s->block_noflag = std::atoi(e) != 0;
    if (const char *e = std::getenv("NMPC_AS_BUILD")) {
        s->as_v256 = std::strcmp(e, "v256") == 0;
        if (std::strcmp(e, "default") == 0) s->as_noflag = 1;
    }
    if (const char *e = std::getenv("NMPC_TEAM_LSTG")) s->team_lstg = std::atoi(e);
    if (const char *e = std::getenv("NMPC_BLOCK_TAIL")) s->block_tail = std::atoi(e) != 0;
    if (const char *e = std::getenv("NMPC_BLOCK_J")) s->block_J = std::atoi(e);
    if (const char *e = std::getenv("NMPC_TAIL_CAP")) s->tail_cap = std::atoi(e);
    if (const char *e = std::getenv("NMPC_TAIL_FWD")) s->tail_fwd = std::atoi(e) != 0;
    if (const char *e = std::getenv("NMPC_TAIL_KEEP")) s->tail_keep = std::atoi(e) != 0;
    if (const char *e = std::getenv("NMPC_TAIL_FWD_OVERLAP")) s->tail_fwd_overlap = std::atoi(e) != 0;
    if (const char *e = std::getenv("NMPC_TEAM_TPW")) {
        const int v = std::atoi(e);
        if (v == 1 || v == 2 || v == 4) s->team_tpw = v;
    }
    s->Bp = (cfg->max_batch + 63) / 64 * 64;
    if (alloc_ws(s) != 0) {
        g_create_error = s->err;
        nmpc_destroy(s);
        return nullptr;
    }
    {
        // block-parallel tail: the common continuation of a long-horizon work-list instance (one warm interior-point iteration, one
        // more attempt) with every factorisation cut into blocks.  Needs the FP64 tile kernels, the warm start, and an attempt
        // schedule in which the second attempt is the last (the tail never leaves a second warm start behind).
        const nmpc_config &g = s->cfg;
        const bool can = (g.flags & NMPC_FLAG_TEAM_MAPPING) && !(g.flags & NMPC_FLAG_CONDENSED_QP) && g.qp_polish &&
                         g.qp_warm_start && g.qp_polish_budget >= g.qp_polish_passes && g.qp_polish_budget <= 2 * g.qp_polish_passes &&
                         g.sim_num_steps <= AS_MAX_STEPS && s->team_split && g.N >= 8;
        const bool want = s->block_tail < 0 ? g.N >= 160 : s->block_tail != 0;
        if (can && want) {
            int J = s->block_J > 0 ? s->block_J : (int)std::lround(0.7 * std::sqrt((double)g.N));
            J = std::max(2, std::min(J, g.N / 2));
            const int M = (g.N + J - 1) / J;
            J = (g.N + M - 1) / M;
            const size_t Bw = (size_t)s->Bp + 1;
            bool ok = gmalloc(s, (void **)&s->d_ts, Bw * TS_ROWS * sizeof(double)) == hipSuccess &&
                      gmalloc(s, (void **)&s->d_binfo, Bw * J * 2 * sizeof(double)) == hipSuccess &&
                      gmalloc(s, (void **)&s->tail_agg, Bw * J * 3 * BLK_MAT * sizeof(double)) == hipSuccess &&
                      gmalloc(s, (void **)&s->tail_bnd, Bw * (J + 1) * BLK_MAT * sizeof(double)) == hipSuccess &&
                      gmalloc(s, (void **)&s->d_wl2, ((size_t)s->Bp + 2) * sizeof(int)) == hipSuccess &&
                      gmalloc(s, (void **)&s->d_wl3, ((size_t)s->Bp + 2) * sizeof(int)) == hipSuccess &&
                      gmalloc(s, (void **)&s->tail_gbuf, Bw * J * BLK_GB * sizeof(double)) == hipSuccess &&
                      gmalloc(s, (void **)&s->tail_xb, Bw * J * 16 * sizeof(double)) == hipSuccess &&
                      gmalloc(s, (void **)&s->tail_frec, Bw * J * FR_ROWS * sizeof(double)) == hipSuccess;
            ok = ok && hipMemset(s->d_wl2, 0, ((size_t)s->Bp + 2) * sizeof(int)) == hipSuccess &&
                 hipMemset(s->d_wl3, 0, ((size_t)s->Bp + 2) * sizeof(int)) == hipSuccess &&
                 hipMemset(s->d_ts, 0, Bw * TS_ROWS * sizeof(double)) == hipSuccess &&
                 hipMemset(s->d_binfo, 0, Bw * J * 2 * sizeof(double)) == hipSuccess;
            if (!ok) {
                g_create_error = "hipMalloc of the block-parallel tail's buffers failed";
                nmpc_destroy(s);
                return nullptr;
            }
            s->tail_J = J; s->tail_M = M;
            s->ws_bytes += (Bw * TS_ROWS + Bw * J * 2 + Bw * J * 3 * BLK_MAT + Bw * (J + 1) * BLK_MAT) * sizeof(double);
        }
    }
#ifdef NMPC_PROFILE
    if (gmalloc(s, (void **)&s->d_prof, (size_t)8 * s->Bp * sizeof(long long)) != hipSuccess) s->d_prof = nullptr;
#endif
    for (auto &e : s->ev)
        if (hipEventCreate(&e) != hipSuccess) { g_create_error = "hipEventCreate failed"; nmpc_destroy(s); return nullptr; }
    const int N = cfg->N;
    s->sx.assign((size_t)(N + 1) * NX, 0.0);
    s->su.assign((size_t)N * NU, 0.0);
    s->syref.assign((size_t)N * NY, 0.0);   // ocp.cost.yref = zeros, controller.py:244-245
    s->syref_e.assign(NX, 0.0);
    s->sx0.assign(NX, 0.0);                 // lbx_0 = ubx_0 = zeros, controller.py:254-255
    return s;
}

void nmpc_destroy(nmpc_solver *s)
{
    if (!s) return;
    (void)hipSetDevice(s->cfg.device);
    (void)hipDeviceSynchronize();
    void *ptrs[] = {s->d_gbase, s->d_consts, s->d_wl, s->d_npol, s->cond, s->tAB, s->tP, s->d_prof, s->AB, s->bv, s->qr, s->xl, s->ul, s->LM, s->iv, s->d_iters, s->d_status, s->s_x0,
                    s->s_yref, s->s_yref_e, s->s_xi, s->s_ui, s->s_u0, s->s_xo, s->s_uo};
    for (void *p : ptrs)
        if (p) (void)gfree(s, p);
    for (void *p : {(void *)s->blk_agg, (void *)s->blk_bnd, (void *)s->blk_chk, (void *)s->blk_fac, (void *)s->d_ts, (void *)s->d_binfo,
                    (void *)s->tail_agg, (void *)s->tail_bnd, (void *)s->d_wl2, (void *)s->d_wl3, (void *)s->tail_gbuf, (void *)s->tail_xb,
                    (void *)s->tail_frec})
        if (p) (void)gfree(s, p);
    for (auto &e : s->blk_ev)
        if (e) (void)hipEventDestroy(e);
    if (s->d_in) (void)gfree(s, s->d_in);
    if (s->d_out) (void)gfree(s, s->d_out);
    if (s->h_in) (void)hipHostFree(s->h_in);
    if (s->h_out) (void)hipHostFree(s->h_out);
    if (s->pack_stream) (void)hipStreamDestroy(s->pack_stream);
    for (auto &e : s->ev)
        if (e) (void)hipEventDestroy(e);
    delete s;
}

const char *nmpc_last_error(const nmpc_solver *s) { return s ? s->err.c_str() : g_create_error.c_str(); }

int nmpc_debug_last_schedule(const nmpc_solver *s)
{
    if (!s) return NMPC_EARG;
    return (s->last_split ? 1 : 0) | ((s->last_split && s->last_inplace) ? 2 : 0) | ((s->last_split && s->last_tail) ? 4 : 0);
}

// Diagnostics (include/rotors_nmpc.h): bytes of the canary bands around the handle's device buffers that no longer hold the fill
// pattern; -1 when the handle was created without NMPC_GUARD.  nmpc_last_error() names the first damaged buffer.
long long nmpc_debug_guard_check(nmpc_solver *s)
{
    if (!s) return NMPC_EARG;
    if (!s->guard) return -1;
    if (hipSetDevice(s->cfg.device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return s->fail(NMPC_EHIP, "guard_check: device synchronisation failed");
    std::vector<unsigned char> h(2 * s->guard);
    long long bad = 0;
    for (size_t i = 0; i < s->guards.size(); i++) {
        const auto &g = s->guards[i];
        if (hipMemcpy(h.data(), g.base, s->guard, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(h.data() + s->guard, g.base + s->guard + g.n, s->guard, hipMemcpyDeviceToHost) != hipSuccess)
            return s->fail(NMPC_EHIP, "guard_check: copy failed");
        long long b = 0, first = -1;
        for (size_t k = 0; k < h.size(); k++)
            if (h[k] != 0xA5) { b++; if (first < 0) first = (long long)k; }
        if (b && !bad)
            s->fail(0, "guard_check: buffer %zu (%zu bytes): %lld damaged canary bytes, first %s the buffer at distance %lld", i, g.n, b,
                    first < (long long)s->guard ? "below" : "above", first < (long long)s->guard ? (long long)s->guard - first : first - (long long)s->guard);
        bad += b;
    }
    return bad;
}

}  // extern "C"

// working arrays of a team in front of its stage cache, in doubles (more than two integrator steps: the evaluation points of
// the shared linearisation's single stage take the larger layout, EvLayout<AS_MAX_STEPS>)
static int as_lds_base(const nmpc_solver *s, bool shared)
{
    return shared ? TEAM_AS_LDS_SHARED + (s->cfg.sim_num_steps > 2 ? AS_EV : 0) : TEAM_AS_LDS_STAGE;
}

// the kernels that iterate the interior point method: the flag build (nmpc_qpf.hip) for what it is validated on - at most two integrator
// steps, as k_team_as - else (and with NMPC_QP_NOFLAG=1) the default code generation (nmpc_qp.hip)
template <class TI>
static int launch_qp_kind(const nmpc_solver *s, const AsLaunch &a, const Inputs<TI> &in, const Outputs<TI> &out)
{
    if (a.kind >= 1 && a.kind <= 3 && s->cfg.sim_num_steps <= 2 && !s->qp_noflag) return launch_team_qp_flag(a, in, out);
    return launch_team_qp(a, in, out);
}

// Where a team's stage cache starts.  The evaluation-point buffer of the linearisation (per-stage variant: 448 of a team's 1280 doubles)
// is dead once the preparation has run, so the cache of the per-stage variant starts AT it (A_EV) instead of behind it: five more stages
// of factors stay in LDS (11 instead of 6 in k_team_as).  The shared variant's single stage of evaluation points is small and stays.
// NMPC_LDS_OVERLAP=0 restores the round-3 placement.
static int as_cache_base(const nmpc_solver *s, bool shared)
{
    return (shared || !s->lds_overlap) ? as_lds_base(s, shared) : A_EV;
}
// stride of a team's LDS (carve + cache), 192 B past a multiple of the 256-B bank row; at least the carve the preparation needs
static int as_lds_stride(const nmpc_solver *s, bool shared, int lstg, int rows)
{
    int stride = std::max(as_lds_base(s, shared), as_cache_base(s, shared) + lstg * rows);
    stride += (s->lds_pad - stride % 32 + 32) % 32;          // (NMPC_LDS_PAD: residue of the team stride mod 32 doubles; 24 = 192 B past a bank row)
    return stride;
}

// LDS carve of the kernels that iterate the interior point method (k_team_qp, k_team_qp_list: one wave per SIMD, 40 KB per wave):
// stage cache rows of IP_LM_ROWS doubles
static void qp_lds(const nmpc_solver *s, bool shared, AsLaunch &a)
{
    const int base = as_cache_base(s, shared);
    const int per_team = 40960 / 4 / (int)sizeof(double);
    int lstg = std::max(0, std::min(s->cfg.N, (per_team - base - 31) / IP_LM_ROWS));
    if (s->team_lstg >= 0) lstg = std::min(lstg, s->team_lstg);
    const int stride = as_lds_stride(s, shared, lstg, IP_LM_ROWS);
    a.lstg = lstg; a.lds_stride = stride; a.lm_off = base; a.lds_bytes = (size_t)4 * stride * sizeof(double);
}

// Default FP64 path: the active-set kernel makes the first attempt of every instance; the general kernel runs on the work
// list of what that attempt could not settle.  TI = element type of the caller's device arrays (double, or float for
// NMPC_DTYPE_F32IO); arithmetic and workspace are double either way.
template <class TI>
static int launch_split(nmpc_solver *s, const Consts<double> &c, const Work<double> &w, const Inputs<TI> &in, const Outputs<TI> &out,
                        const TeamWork<double> &tw, int B, int tpw, hipStream_t st)
{
    const dim3 tgrid((B + tpw - 1) / tpw);
    WorkList wl;
    wl.count = s->d_wl; wl.done = s->d_wl + 1; wl.list = s->d_wl + 2;
    const int nlist = std::min((B + 3) / 4, s->list_grid);      // usually empty: keep the launch small (each workgroup strides over the list)
    const bool traj = out.x_out != nullptr || out.u_out != nullptr;
    int occ_as = s->team_occ;
    // Two waves per SIMD (256 registers, no LDS stage cache) pay only where one wave per SIMD would leave a small second round: just
    // above 1024 waves.  Re-measured on the round-4 builds (profiles/r04v_occupancy_sweep.txt; two-wave / one-wave build, M solves/s):
    // B = 4608 51.8 / 49.4, B = 5120 56.7 / 54.5, B = 6144 49.7 / 51.4, B = 7168 67.8 / 82.0, B = 8192 57.8 / 67.4, B = 16384 69.5 / 82.8,
    // B = 65536 84.4 / 94.9 - the one-wave build gained more from the scheduler strategy, and the range that was (1024, 2048] waves until
    // round 4 cost 15-20 % at its upper end.  The per-stage variant spills at 256 registers and never takes the two-wave build.
    const int nwaves = (B + tpw - 1) / tpw;
    if (occ_as == 0) occ_as = (nwaves > 1024 && nwaves <= 1408) ? 2 : 1;
    if (!c.shared) occ_as = 1;
    // LDS stage cache: what is left of the CU's 160 KB at this occupancy (40 KB per wave at one wave per SIMD) holds the
    // factors of the first stages; the team stride stays 192 B past a multiple of the 256-B bank row (24 doubles mod 32).
    // The two-waves build carries no cache.
    const int base_as = as_cache_base(s, c.shared != 0);
    const int per_team = 40960 / 4 / (int)sizeof(double);
    int lstg = occ_as == 2 ? 0 : std::max(0, std::min(s->cfg.N, (per_team - base_as - 31) / AS_LM_ROWS));
    if (s->team_lstg >= 0) lstg = std::min(lstg, s->team_lstg);
    const int lds_stride = as_lds_stride(s, c.shared != 0, lstg, AS_LM_ROWS);
    const size_t lds_as = (size_t)4 * lds_stride * sizeof(double);
    AsLaunch al;
    al.cp = (const Consts<double> *)s->d_consts; al.w = w; al.tw = tw; al.wl = wl; al.B = B; al.tpw = tpw;
    al.lds_stride = lds_stride; al.lstg = lstg; al.lm_off = base_as; al.occ = occ_as; al.shared = c.shared != 0; al.traj = traj;
    al.lds_bytes = lds_as; al.stream = st;
    const bool tail = s->tail_J > 0;
    // failed first attempts continue inside k_team_as (team_as_kernel): the one-wave builds, the team_as kernels, short horizons
    const bool inplace = s->team_inplace && !tail && occ_as == 1 && !s->as_v256 && as_cont_built(c.shared != 0, traj);
    if (inplace) {
        AsLaunch ql = al;
        qp_lds(s, c.shared != 0, ql);               // the MODE 2 carve: IP_LM_ROWS per cached stage, same base
        al.cont_stride = ql.lds_stride; al.cont_lstg = ql.lstg;
        al.lds_bytes = std::max(al.lds_bytes, ql.lds_bytes);
    }
    const int cap = tail ? std::max(1, std::min(s->tail_cap > 0 ? s->tail_cap : s->cfg.qp_polish_passes, s->cfg.qp_polish_passes)) : 0;
    if (tail) {
        // the tail's lists are reset by its last launches; a solve that returned early with an error leaves counts behind, and the next
        // one would append past list[Bp]: the three headers (count | done) start every solve at zero
        for (int *h : {s->d_wl, s->d_wl2, s->d_wl3}) HIP_TRY(s, hipMemsetAsync(h, 0, 2 * sizeof(int), st));
        HIP_TRY(s, hipMemsetAsync(s->d_ts, 0, ((size_t)s->Bp + 1) * TS_ROWS * sizeof(double), st));
        al.tail.cap = cap < s->cfg.qp_polish_passes ? cap : 0;
        al.tail.ts = s->d_ts;
    }
    // k_team_as: the flag build (nmpc_as.hip) for what it is validated on, the default-codegen build (nmpc_qp.hip) otherwise
    if (s->as_v256 && al.shared && al.occ == 1) { al.occ = 3; HIP_TRY(s, (hipError_t)launch_team_qp(al, in, out)); al.occ = 1; }
    else if (s->cfg.sim_num_steps <= 2 && !s->as_noflag) HIP_TRY(s, (hipError_t)launch_team_as(al, in, out));
    else HIP_TRY(s, (hipError_t)launch_team_qp(al, in, out));
    if (s->timing) HIP_TRY(s, hipEventRecord(s->ev[2], st));
    if (tail) {
        // long horizon: the work list continues in steps - the factorisation of a step by the block-parallel launches of
        // nmpc_block.hip (J blocks of the horizon at the same time), the rest of the iteration / pass by k_team_tail - and
        // what leaves the common path is solved by k_team_qp_list from the hand-over, exactly as without the tail
        WorkList wl2;
        wl2.count = s->d_wl2; wl2.done = s->d_wl2 + 1; wl2.list = s->d_wl2 + 2;
        const int ngrid = std::min((B + 3) / 4, 256);
        AsLaunch tl = al;
        tl.kind = 3; tl.nlist = ngrid; tl.tpw = 4; tl.occ = 1; tl.lstg = 0;
        tl.lm_off = as_lds_base(s, c.shared != 0);
        tl.lds_stride = tl.lm_off + (24 - tl.lm_off % 32 + 32) % 32;
        tl.lds_bytes = (size_t)4 * tl.lds_stride * sizeof(double);
        tl.tail.ts = s->d_ts; tl.tail.binfo = s->d_binfo; tl.tail.J = s->tail_J; tl.tail.fb_count = wl2.count; tl.tail.fb_list = wl2.list;
        BlockLaunch bl{};
        bl.cp = al.cp;
        bl.g.tAB = (const double *)s->tAB; bl.g.tIV = (const double *)s->iv;
        bl.g.agg = s->tail_agg; bl.g.bnd = s->tail_bnd; bl.g.bchk = nullptr; bl.g.fac = nullptr;
        bl.g.Bp = s->Bp; bl.g.B = B; bl.g.J = s->tail_J; bl.g.M = s->tail_M;
        bl.g.list = wl.list; bl.g.count = wl.count; bl.g.ts = s->d_ts; bl.g.tLM = (double *)s->LM; bl.g.binfo = s->d_binfo;
        bl.g.shared = c.shared != 0;
        bl.g.gbuf = s->tail_fwd ? s->tail_gbuf : nullptr; bl.g.xb = s->tail_xb;
        bl.g.frec = (s->tail_fwd && s->tail_keep) ? s->tail_frec : nullptr;
        bl.g.fwd_in_sweep = s->tail_fwd_overlap;
        bl.stream = st; bl.timing = false; bl.tail_grid = ngrid;
        for (auto &e : bl.ev) e = nullptr;
        // the work list is compacted from step to step: every step reads one list and appends what is still in the tail to the other
        WorkList lists[2];
        lists[0] = wl;
        lists[1].count = s->d_wl3; lists[1].done = s->d_wl3 + 1; lists[1].list = s->d_wl3 + 2;
        int cur = 0;
        auto step = [&](int phase, bool factor) -> int {
            tl.wl = lists[cur]; tl.tail.nx_count = lists[cur ^ 1].count; tl.tail.nx_list = lists[cur ^ 1].list;
            tl.tail.phase = phase;
            if (factor) {
                bl.g.list = lists[cur].list; bl.g.count = lists[cur].count; bl.g.reset_count = nullptr;
                // the list this step appends to was consumed by the previous step: its count is zeroed between the two block sweeps
                bl.g.reset_count = lists[cur ^ 1].count;
                HIP_TRY(s, (hipError_t)(s->block_noflag ? launch_block_factor(bl, in) : launch_block_factor_flag(bl, in)));
            }
            tl.tail.frec = nullptr;
            if (phase == 2 && s->tail_fwd) {
                // forward sweep of the pass: the blocks of every instance at the same time (phase 3), then the decision (phase 2)
                tl.tail.frec = s->tail_frec; tl.tail.xb = s->tail_xb; tl.tail.M = s->tail_M;
                tl.tail.phase = 3;
                HIP_TRY(s, (hipError_t)launch_qp_kind(s, tl, in, out));
                tl.tail.phase = 2;
            }
            HIP_TRY(s, (hipError_t)launch_qp_kind(s, tl, in, out));
            cur ^= 1;
            return 0;
        };
        if (step(0, false)) return NMPC_EHIP;
        for (int p = cap; p < s->cfg.qp_polish_passes; p++)          // the rest of the first attempt
            if (step(2, true)) return NMPC_EHIP;
        if (s->cfg.qp_polish_budget > s->cfg.qp_polish_passes) {     // (a one-attempt policy - long horizons by default - ends here: what is left goes to the fallback list)
            if (step(1, true)) return NMPC_EHIP;                     // one interior-point iteration from the warm start
            for (int p = 0; p < s->cfg.qp_polish_budget - s->cfg.qp_polish_passes; p++)   // the second attempt
                if (step(2, true)) return NMPC_EHIP;
        }
        AsLaunch ql = al;
        qp_lds(s, c.shared != 0, ql);
        ql.kind = 2; ql.nlist = nlist; ql.tpw = 4; ql.occ = 1; ql.wl = wl2;
        HIP_TRY(s, (hipError_t)launch_qp_kind(s, ql, in, out));
        AsLaunch rl = al;
        rl.kind = 4; rl.tail.nx_count = lists[1].count;
        HIP_TRY(s, (hipError_t)launch_team_qp(rl, in, out));
    } else if (inplace) {
        // nothing was handed over: every instance finished inside k_team_as
    } else {
        AsLaunch ql = al;
        qp_lds(s, c.shared != 0, ql);
        ql.kind = 2; ql.nlist = nlist; ql.tpw = 4; ql.occ = 1;
        HIP_TRY(s, (hipError_t)launch_qp_kind(s, ql, in, out));
    }
    HIP_TRY(s, hipGetLastError());
    if (s->timing) HIP_TRY(s, hipEventRecord(s->ev[3], st));
    s->last_B = B; s->solved = true; s->timed = s->timing; s->timed_fused = true; s->timed_split = true; s->last_split = true;
    s->last_inplace = inplace; s->last_tail = tail;
    s->last_shared = c.shared != 0;
    return 0;
}

// One solve.  TI = element type of the caller's device arrays (double; float for NMPC_DTYPE_F32IO / _F32): arithmetic and workspace are
// double either way.  The default path (launch_split), the one-launch interior-point kernel k_team_qp for every other team-mapped
// configuration, and - FP64 arrays only - the two fidelity paths of the lane layout (k_prepare + k_ipm | k_cond_ipm)
template <class TI>
static int launch_any(nmpc_solver *s, int B, const void *x0, const void *yref, const void *yref_e, int bcast,
                      const void *x_init, const void *u_init, void *u0, int32_t *status, void *x_out, void *u_out, hipStream_t st)
{
    using T = double;
    Consts<T> c;
    fill_consts(s->cfg, c);
    const bool cold = (x_init == nullptr || u_init == nullptr);
    c.shared = (cold && (s->cfg.flags & NMPC_FLAG_SHARE_COLD_START)) ? 1 : 0;
    Work<T> w;
    w.Bp = s->Bp;
    w.AB = (T *)s->AB; w.bv = (T *)s->bv; w.qr = (T *)s->qr; w.xl = (T *)s->xl; w.ul = (T *)s->ul;
    w.LM = (T *)s->LM; w.iv = (T *)s->iv; w.iters = s->d_iters; w.status = s->d_status;
    w.prof = s->d_prof;
    w.npol = s->d_npol;
    w.tAB = (s->cfg.flags & NMPC_FLAG_TEAM_MAPPING) ? (T *)s->tAB : nullptr;
    w.gbase = (T *)s->d_gbase;
    Inputs<TI> in;
    in.x0 = (const TI *)x0; in.yref = (const TI *)yref; in.yref_e = (const TI *)yref_e;
    in.x_init = cold ? nullptr : (const TI *)x_init; in.u_init = cold ? nullptr : (const TI *)u_init;
    in.yref_bcast = bcast;
    Outputs<TI> out;
    out.u0 = (TI *)u0; out.x_out = (TI *)x_out; out.u_out = (TI *)u_out; out.status = status;
    const dim3 grid((B + 63) / 64), block(64);
    if (s->timing) HIP_TRY(s, hipEventRecord(s->ev[0], st));
    const bool team = (s->cfg.flags & NMPC_FLAG_TEAM_MAPPING) && !(s->cfg.flags & NMPC_FLAG_CONDENSED_QP);
    if (team) {
        // the team kernels linearise inside the solve kernel and keep the stage data as [inst][rows]
        TeamWork<T> tw;
        tw.tLM = (T *)s->LM;
        tw.tIV = (T *)s->iv;
        tw.tP = (s->cfg.qp_polish && c.polish_ckpt > 0) ? (T *)s->tP : nullptr;
        // teams per wave: 4 fills the lanes; fewer (half-empty waves) when the batch alone cannot
        // put two waves on every SIMD, so that LDS/memory latency still has something to hide behind
        int tpw = s->team_tpw;
        if (tpw == 0) tpw = (B >= 2048) ? 4 : (B >= 512 ? 2 : 1);   // measured: 4 is best from B = 4096 up
        // Default path: the active-set kernel makes the first attempt of every instance and continues what that attempt cannot
        // settle.  Taken when the QP starts with an active-set attempt (polish on, first attempt before any interior-point iteration).
        const bool split = s->team_split && s->cfg.qp_polish && s->cfg.qp_polish_budget > 0 &&
                           s->cfg.qp_polish_passes > 0 && s->cfg.qp_polish_mu >= s->cfg.qp_mu0 && !(s->cfg.qp_mu0 <= s->cfg.qp_tol_comp);
        if (split) return launch_split<TI>(s, c, w, in, out, tw, B, tpw, st);
        // every other tile-form solve - qp_polish = 0, an attempt schedule that starts with interior-point iterations,
        // NMPC_TEAM_SPLIT=0 - is ONE launch of k_team_qp (the same sweeps as the split path, nmpc_team_as.hpp)
        WorkList wl;
        wl.count = s->d_wl; wl.done = s->d_wl + 1; wl.list = s->d_wl + 2;
        AsLaunch al;
        al.cp = (const Consts<double> *)s->d_consts; al.w = w; al.tw = tw; al.wl = wl; al.B = B; al.tpw = tpw;
        al.occ = 1; al.shared = c.shared != 0; al.traj = out.x_out != nullptr || out.u_out != nullptr;
        al.stream = st; al.kind = 1;
        qp_lds(s, c.shared != 0, al);
        // the flag build (nmpc_qpf.hip) for what it is validated on - at most two integrator steps, as k_team_as - else the default one
        HIP_TRY(s, (hipError_t)launch_qp_kind(s, al, in, out));
        if (s->timing) HIP_TRY(s, hipEventRecord(s->ev[2], st));
        s->last_B = B; s->solved = true; s->timed = s->timing; s->timed_fused = true; s->timed_split = false; s->last_split = false;
        s->last_shared = c.shared != 0;
        return 0;
    }
    // lane layout ([row][Bp] workspace; FP64 arrays only, checked at create): linearisation as a launch of its own, then the QP
    if constexpr (!std::is_same<TI, double>::value) {
        return s->fail(NMPC_EARG, "solve_batch: FP32 buffers need the team mapping without condensing");
    } else {
    HIP_TRY(s, hipMemsetAsync(s->d_npol, 0, (size_t)B * sizeof(int32_t), st));   // (the lane kernels make no active-set passes)
    hipLaunchKernelGGL(k_prepare<T>, grid, block, 0, st, c, w, in, B);
    HIP_TRY(s, hipGetLastError());
    if (s->timing) HIP_TRY(s, hipEventRecord(s->ev[1], st));
    if (s->cfg.flags & NMPC_FLAG_CONDENSED_QP) {
        CondWork<T> cw;
        const int N2 = (s->cfg.qp_cond_N > 0 && s->cfg.qp_cond_N < s->cfg.N) ? s->cfg.qp_cond_N : s->cfg.N;
        cond_layout(cw, s->cfg.N, N2);
        cw.base = (T *)s->cond;
        cw.Bp = s->Bp;
        hipLaunchKernelGGL(k_cond_ipm<T>, grid, block, 0, st, c, w, cw, in, out, B);
    } else {
        hipLaunchKernelGGL(k_ipm<T>, grid, block, 0, st, c, w, in, out, B);
    }
    HIP_TRY(s, hipGetLastError());
    if (s->timing) HIP_TRY(s, hipEventRecord(s->ev[2], st));
    s->last_B = B;
    s->solved = true;
    s->timed = s->timing;
    s->timed_fused = false;
    s->timed_split = false;
    s->last_split = false;
    s->last_shared = c.shared != 0;
    return 0;
    }
}

extern "C" {

int nmpc_solve_batch_device(nmpc_solver *s, int B, const void *x0, const void *yref, const void *yref_e,
                            int yref_bcast, const void *x_init, const void *u_init, void *u0,
                            int32_t *status, void *x_out, void *u_out, void *hip_stream)
{
    if (!s) return NMPC_EARG;
    if (B < 1 || B > s->cfg.max_batch) return s->fail(NMPC_EARG, "solve_batch: B=%d outside [1, max_batch=%d]", B, s->cfg.max_batch);
    if (!x0 || !yref || !yref_e || !u0) return s->fail(NMPC_EARG, "solve_batch: x0, yref, yref_e and u0 are required");
    if ((x_init == nullptr) != (u_init == nullptr)) return s->fail(NMPC_EARG, "solve_batch: x_init and u_init must both be given or both be NULL");
    HIP_TRY(s, hipSetDevice(s->cfg.device));
    hipStream_t st = (hipStream_t)hip_stream;
    if (s->cfg.dtype == NMPC_DTYPE_F32IO) return launch_any<float>(s, B, x0, yref, yref_e, yref_bcast, x_init, u_init, u0, status, x_out, u_out, st);
    return launch_any<double>(s, B, x0, yref, yref_e, yref_bcast, x_init, u_init, u0, status, x_out, u_out, st);
}

static int ensure_staging(nmpc_solver *s, size_t B)
{
    if (B <= s->staged_batch) return 0;
    s->staged_batch = 0;      // a failed grow below must force a fresh allocation on the next call
    void **ps[] = {&s->s_x0, &s->s_yref, &s->s_yref_e, &s->s_xi, &s->s_ui, &s->s_u0, &s->s_xo, &s->s_uo};
    for (void **p : ps)
        if (*p) { (void)gfree(s, *p); *p = nullptr; }
    const size_t N = (size_t)s->cfg.N, e = s->esz;
    HIP_TRY(s, gmalloc(s, &s->s_x0, B * NX * e));
    HIP_TRY(s, gmalloc(s, &s->s_yref, B * N * NY * e));
    HIP_TRY(s, gmalloc(s, &s->s_yref_e, B * NX * e));
    HIP_TRY(s, gmalloc(s, &s->s_xi, B * (N + 1) * NX * e));
    HIP_TRY(s, gmalloc(s, &s->s_ui, B * N * NU * e));
    HIP_TRY(s, gmalloc(s, &s->s_u0, B * NU * e));
    HIP_TRY(s, gmalloc(s, &s->s_xo, B * (N + 1) * NX * e));
    HIP_TRY(s, gmalloc(s, &s->s_uo, B * N * NU * e));
    s->staged_batch = B;
    return 0;
}

static int h2d(nmpc_solver *s, void *dst, const double *src, size_t n)
{
    if (s->esz == 8) {
        HIP_TRY(s, hipMemcpy(dst, src, n * 8, hipMemcpyHostToDevice));
    } else {
        if (s->cvt.size() < n) s->cvt.resize(n);         // persistent conversion buffer: grows, never shrinks
        for (size_t i = 0; i < n; i++) s->cvt[i] = (float)src[i];
        HIP_TRY(s, hipMemcpy(dst, s->cvt.data(), n * 4, hipMemcpyHostToDevice));
    }
    return 0;
}

static int d2h(nmpc_solver *s, double *dst, const void *src, size_t n)
{
    if (s->esz == 8) {
        HIP_TRY(s, hipMemcpy(dst, src, n * 8, hipMemcpyDeviceToHost));
    } else {
        if (s->cvt.size() < n) s->cvt.resize(n);
        HIP_TRY(s, hipMemcpy(s->cvt.data(), src, n * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; i++) dst[i] = (double)s->cvt[i];
    }
    return 0;
}

// host doubles -> element type of the device buffers, into (pinned) staging memory
static void pack_put(size_t esz, void *dst, const double *src, size_t n)
{
    if (esz == 8) std::memcpy(dst, src, n * 8);
    else { float *d = (float *)dst; for (size_t i = 0; i < n; i++) d[i] = (float)src[i]; }
}
static void pack_get(size_t esz, double *dst, const void *src, size_t n)
{
    if (esz == 8) std::memcpy(dst, src, n * 8);
    else { const float *f = (const float *)src; for (size_t i = 0; i < n; i++) dst[i] = (double)f[i]; }
}

// Latency path (B <= PACK_B; config 1 of BASELINE.json is B = 1 at 60 Hz, config/params.yaml:48): everything
// the solve reads travels in one pinned block and one H2D copy, everything it writes in one D2H copy, all on
// the solver's own stream; one stream synchronisation at the end.
static int solve_packed(nmpc_solver *s, int B, const double *x0, const double *yref, const double *yref_e,
                        int yref_bcast, const double *x_init, const double *u_init, double *u0,
                        int32_t *status, double *x_out, double *u_out)
{
    const size_t N = (size_t)s->cfg.N, e = s->esz, P = nmpc_solver::PACK_B;
    if (!s->pack_ready) {
        // all five resources or none: a partial failure frees what it got, so that the next call starts over
        const size_t nin = P * (NX + N * NY + NX + (N + 1) * NX + N * NU), nout = P * (NU + (N + 1) * NX + N * NU);
        s->pack_in_bytes = nin * e;
        s->pack_out_bytes = nout * e + P * sizeof(int32_t) + 8;
        hipError_t pe = hipStreamCreateWithFlags(&s->pack_stream, hipStreamNonBlocking);
        if (pe == hipSuccess) pe = hipHostMalloc(&s->h_in, s->pack_in_bytes, hipHostMallocDefault);
        if (pe == hipSuccess) pe = hipHostMalloc(&s->h_out, s->pack_out_bytes, hipHostMallocDefault);
        if (pe == hipSuccess) pe = gmalloc(s, &s->d_in, s->pack_in_bytes);
        if (pe == hipSuccess) pe = gmalloc(s, &s->d_out, s->pack_out_bytes);
        if (pe != hipSuccess) {
            if (s->d_out) (void)gfree(s, s->d_out);
            if (s->d_in) (void)gfree(s, s->d_in);
            if (s->h_out) (void)hipHostFree(s->h_out);
            if (s->h_in) (void)hipHostFree(s->h_in);
            if (s->pack_stream) (void)hipStreamDestroy(s->pack_stream);
            s->d_out = s->d_in = s->h_out = s->h_in = nullptr; s->pack_stream = nullptr;
            return s->fail(NMPC_EHIP, "latency path: allocation failed: %s", hipGetErrorString(pe));
        }
        s->pack_ready = true;
    }
    // the workspace is shared with nmpc_solve_batch_device: work the caller enqueued on other streams must have drained before
    // this stream touches it (the header asks the caller to synchronise; this makes the common case - everything on the NULL
    // stream or on one stream that was synchronised - safe without it)
    HIP_TRY(s, hipStreamSynchronize(nullptr));
    const size_t Bs = (size_t)B, nyr = (yref_bcast ? 1 : Bs) * N * NY, nye = (yref_bcast ? 1 : Bs) * NX;
    const bool warm = x_init != nullptr;
    const size_t o_x0 = 0, o_yr = o_x0 + Bs * NX, o_ye = o_yr + nyr, o_xi = o_ye + nye, o_ui = o_xi + (warm ? Bs * (N + 1) * NX : 0),
                 n_in = o_ui + (warm ? Bs * N * NU : 0);
    char *hi = (char *)s->h_in, *di = (char *)s->d_in, *ho = (char *)s->h_out, *dO = (char *)s->d_out;
    pack_put(e, hi + o_x0 * e, x0, Bs * NX);
    pack_put(e, hi + o_yr * e, yref, nyr);
    pack_put(e, hi + o_ye * e, yref_e, nye);
    if (warm) { pack_put(e, hi + o_xi * e, x_init, Bs * (N + 1) * NX); pack_put(e, hi + o_ui * e, u_init, Bs * N * NU); }
    const size_t q_u0 = 0, q_xo = q_u0 + Bs * NU, q_uo = q_xo + (x_out ? Bs * (N + 1) * NX : 0), n_out = q_uo + (u_out ? Bs * N * NU : 0);
    const size_t q_st = (n_out * e + 7) / 8 * 8;                      // int32 status block, 8-byte aligned
    hipStream_t st = s->pack_stream;
    HIP_TRY(s, hipMemcpyAsync(di, hi, n_in * e, hipMemcpyHostToDevice, st));
    const int rc = nmpc_solve_batch_device(s, B, di + o_x0 * e, di + o_yr * e, di + o_ye * e, yref_bcast,
                                           warm ? di + o_xi * e : nullptr, warm ? di + o_ui * e : nullptr, dO + q_u0 * e,
                                           (int32_t *)(dO + q_st), x_out ? dO + q_xo * e : nullptr,
                                           u_out ? dO + q_uo * e : nullptr, st);
    if (rc) return rc;
    HIP_TRY(s, hipMemcpyAsync(ho, dO, q_st + Bs * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(s, hipStreamSynchronize(st));
    pack_get(e, u0, ho + q_u0 * e, Bs * NU);
    if (x_out) pack_get(e, x_out, ho + q_xo * e, Bs * (N + 1) * NX);
    if (u_out) pack_get(e, u_out, ho + q_uo * e, Bs * N * NU);
    if (status) std::memcpy(status, ho + q_st, Bs * sizeof(int32_t));
    return 0;
}

int nmpc_solve_batch(nmpc_solver *s, int B, const double *x0, const double *yref, const double *yref_e,
                     int yref_bcast, const double *x_init, const double *u_init, double *u0,
                     int32_t *status, double *x_out, double *u_out)
{
    if (!s) return NMPC_EARG;
    if (B < 1 || B > s->cfg.max_batch) return s->fail(NMPC_EARG, "solve_batch: B=%d outside [1, max_batch=%d]", B, s->cfg.max_batch);
    if (!x0 || !yref || !yref_e || !u0) return s->fail(NMPC_EARG, "solve_batch: x0, yref, yref_e and u0 are required");
    if ((x_init == nullptr) != (u_init == nullptr)) return s->fail(NMPC_EARG, "solve_batch: x_init and u_init must both be given or both be NULL");
    HIP_TRY(s, hipSetDevice(s->cfg.device));
    if (B <= nmpc_solver::PACK_B) return solve_packed(s, B, x0, yref, yref_e, yref_bcast, x_init, u_init, u0, status, x_out, u_out);
    int rc = ensure_staging(s, (size_t)B);
    if (rc) return rc;
    const size_t N = (size_t)s->cfg.N, Bs = (size_t)B;
    if ((rc = h2d(s, s->s_x0, x0, Bs * NX))) return rc;
    if ((rc = h2d(s, s->s_yref, yref, (yref_bcast ? 1 : Bs) * N * NY))) return rc;
    if ((rc = h2d(s, s->s_yref_e, yref_e, (yref_bcast ? 1 : Bs) * NX))) return rc;
    if (x_init) {
        if ((rc = h2d(s, s->s_xi, x_init, Bs * (N + 1) * NX))) return rc;
        if ((rc = h2d(s, s->s_ui, u_init, Bs * N * NU))) return rc;
    }
    rc = nmpc_solve_batch_device(s, B, s->s_x0, s->s_yref, s->s_yref_e, yref_bcast, x_init ? s->s_xi : nullptr,
                                 x_init ? s->s_ui : nullptr, s->s_u0, nullptr, x_out ? s->s_xo : nullptr,
                                 u_out ? s->s_uo : nullptr, nullptr);
    if (rc) return rc;
    HIP_TRY(s, hipDeviceSynchronize());
    if ((rc = d2h(s, u0, s->s_u0, Bs * NU))) return rc;
    if (status) HIP_TRY(s, hipMemcpy(status, s->d_status, Bs * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (x_out && (rc = d2h(s, x_out, s->s_xo, Bs * (N + 1) * NX))) return rc;
    if (u_out && (rc = d2h(s, u_out, s->s_uo, Bs * N * NU))) return rc;
    return 0;
}

// ---- single-instance AcadosOcpSolver surface (controller.py:412-460)
static int slot(nmpc_solver *s, int stage, const char *field, int n, double **p, bool for_set)
{
    if (!s) return NMPC_EARG;
    if (!field) return s->fail(NMPC_EARG, "field is NULL");
    const int N = s->cfg.N;
    const std::string f(field);
    if (f == "x") {
        if (stage < 0 || stage > N) return s->fail(NMPC_EARG, "stage %d out of range for 'x' (0..%d)", stage, N);
        if (n != NX) return s->fail(NMPC_EARG, "'x' takes %d values, got %d", NX, n);
        *p = &s->sx[(size_t)stage * NX];
        return 0;
    }
    if (f == "u") {
        if (stage < 0 || stage >= N) return s->fail(NMPC_EARG, "stage %d out of range for 'u' (0..%d)", stage, N - 1);
        if (n != NU) return s->fail(NMPC_EARG, "'u' takes %d values, got %d", NU, n);
        *p = &s->su[(size_t)stage * NU];
        return 0;
    }
    if (!for_set) return s->fail(NMPC_EARG, "get: unknown field '%s' (known: x, u)", field);
    if (f == "yref") {
        if (stage < 0 || stage > N) return s->fail(NMPC_EARG, "stage %d out of range for 'yref' (0..%d)", stage, N);
        const int want = stage < N ? NY : NX;
        if (n != want) return s->fail(NMPC_EARG, "'yref' at stage %d takes %d values, got %d", stage, want, n);
        *p = stage < N ? &s->syref[(size_t)stage * NY] : s->syref_e.data();
        return 0;
    }
    if (f == "lbx" || f == "ubx") {
        if (stage != 0)
            return s->fail(NMPC_EARG, "'%s' is only supported at stage 0 (initial-state pin, controller.py:414-415); "
                                      "path state bounds are the inactive +-1e6 box of controller.py:257-261", field);
        if (n != NX) return s->fail(NMPC_EARG, "'%s' takes %d values, got %d", field, NX, n);
        *p = s->sx0.data();
        return 0;
    }
    return s->fail(NMPC_EARG, "set: unknown field '%s' (known: x, u, yref, lbx, ubx)", field);
}

int nmpc_set(nmpc_solver *s, int stage, const char *field, const double *value, int n)
{
    double *p = nullptr;
    if (!s) return NMPC_EARG;
    if (!value) return s->fail(NMPC_EARG, "set: value is NULL");
    const int rc = slot(s, stage, field, n, &p, true);
    if (rc) return rc;
    std::memcpy(p, value, sizeof(double) * (size_t)n);
    return 0;
}

int nmpc_get(nmpc_solver *s, int stage, const char *field, double *out, int n)
{
    double *p = nullptr;
    if (!s) return NMPC_EARG;
    if (!out) return s->fail(NMPC_EARG, "get: out is NULL");
    const int rc = slot(s, stage, field, n, &p, false);
    if (rc) return rc;
    std::memcpy(out, p, sizeof(double) * (size_t)n);
    return 0;
}

int nmpc_solve(nmpc_solver *s)
{
    if (!s) return NMPC_EARG;
    const int N = s->cfg.N;
    double u0[NU];
    int32_t st = 0;
    if (s->sxo.empty()) { s->sxo.resize((size_t)(N + 1) * NX); s->suo.resize((size_t)N * NU); }
    // the linearisation point is whatever set('x'/'u') left in the slot (controller.py:416-431);
    // the stage-0 state is pinned to lbx_0 = ubx_0 (controller.py:414-415)
    const int rc = nmpc_solve_batch(s, 1, s->sx0.data(), s->syref.data(), s->syref_e.data(), 1, s->sx.data(),
                                    s->su.data(), u0, &st, s->sxo.data(), s->suo.data());
    if (rc) return rc;
    if (st == 0) { s->sx.swap(s->sxo); s->su.swap(s->suo); }
    return (int)st;
}

int nmpc_build_hover_reference_device(nmpc_solver *s, int B, const void *positions, const void *yaws,
                                      double thrust_per_motor, void *yref, void *yref_e, void *hip_stream)
{
    if (!s) return NMPC_EARG;
    if (B < 1 || !positions || !yaws || !yref || !yref_e) return s->fail(NMPC_EARG, "build_hover_reference: bad arguments");
    HIP_TRY(s, hipSetDevice(s->cfg.device));
    hipStream_t st = (hipStream_t)hip_stream;
    const int N = s->cfg.N;
    const size_t n = (size_t)B * N * 17 + (size_t)B * 13;
    const dim3 grid((unsigned)std::min<size_t>((n + 255) / 256, 8192)), block(256);
    if (s->cfg.dtype == NMPC_DTYPE_F64)
        hipLaunchKernelGGL(k_hover_reference<double>, grid, block, 0, st, B, N, (const double *)positions,
                           (const double *)yaws, thrust_per_motor, (double *)yref, (double *)yref_e);
    else
        hipLaunchKernelGGL(k_hover_reference<float>, grid, block, 0, st, B, N, (const float *)positions,
                           (const float *)yaws, (float)thrust_per_motor, (float *)yref, (float *)yref_e);
    HIP_TRY(s, hipGetLastError());
    return 0;
}

int nmpc_odometry_to_state_device(nmpc_solver *s, int B, const void *pose, const void *twist, void *x0,
                                  void *hip_stream)
{
    if (!s) return NMPC_EARG;
    if (B < 1 || !pose || !twist || !x0) return s->fail(NMPC_EARG, "odometry_to_state: bad arguments");
    HIP_TRY(s, hipSetDevice(s->cfg.device));
    hipStream_t st = (hipStream_t)hip_stream;
    const dim3 grid((B + 255) / 256), block(256);
    if (s->cfg.dtype == NMPC_DTYPE_F64)
        hipLaunchKernelGGL(k_odometry_to_state<double>, grid, block, 0, st, B, (const double *)pose, (const double *)twist, (double *)x0);
    else
        hipLaunchKernelGGL(k_odometry_to_state<float>, grid, block, 0, st, B, (const float *)pose, (const float *)twist, (float *)x0);
    HIP_TRY(s, hipGetLastError());
    return 0;
}

int nmpc_commands_to_motor_speeds_device(nmpc_solver *s, int B, const void *u, double kf, double w_min,
                                         double w_max, void *speeds, void *clipped, void *hip_stream)
{
    if (!s) return NMPC_EARG;
    if (B < 1 || !u || !speeds) return s->fail(NMPC_EARG, "commands_to_motor_speeds: bad arguments");
    HIP_TRY(s, hipSetDevice(s->cfg.device));
    hipStream_t st = (hipStream_t)hip_stream;
    const int n = B * NU;
    const dim3 grid((n + 255) / 256), block(256);
    if (s->cfg.dtype == NMPC_DTYPE_F64) {
        Bounds4<double> bd;
        for (int i = 0; i < 4; i++) { bd.lb[i] = s->cfg.lbu[i]; bd.ub[i] = s->cfg.ubu[i]; }
        hipLaunchKernelGGL(k_motor_speeds<double>, grid, block, 0, st, n, (const double *)u, bd, kf, w_min, w_max,
                           (double *)speeds, (double *)clipped);
    } else {
        Bounds4<float> bd;
        for (int i = 0; i < 4; i++) { bd.lb[i] = (float)s->cfg.lbu[i]; bd.ub[i] = (float)s->cfg.ubu[i]; }
        hipLaunchKernelGGL(k_motor_speeds<float>, grid, block, 0, st, n, (const float *)u, bd, (float)kf, (float)w_min,
                           (float)w_max, (float *)speeds, (float *)clipped);
    }
    HIP_TRY(s, hipGetLastError());
    return 0;
}

int nmpc_hold_command_device(nmpc_solver *s, int B, const void *u0, const int32_t *status, void *held, void *hip_stream)
{
    if (!s) return NMPC_EARG;
    if (B < 1 || !u0 || !status || !held) return s->fail(NMPC_EARG, "hold_command: bad arguments");
    HIP_TRY(s, hipSetDevice(s->cfg.device));
    hipStream_t st = (hipStream_t)hip_stream;
    const int n = B * NU;
    const dim3 grid((n + 255) / 256), block(256);
    if (s->cfg.dtype == NMPC_DTYPE_F64) {
        Bounds4<double> bd;
        for (int i = 0; i < 4; i++) { bd.lb[i] = s->cfg.lbu[i]; bd.ub[i] = s->cfg.ubu[i]; }
        hipLaunchKernelGGL(k_hold_command<double>, grid, block, 0, st, n, (const double *)u0, status, bd, (double *)held);
    } else {
        Bounds4<float> bd;
        for (int i = 0; i < 4; i++) { bd.lb[i] = (float)s->cfg.lbu[i]; bd.ub[i] = (float)s->cfg.ubu[i]; }
        hipLaunchKernelGGL(k_hold_command<float>, grid, block, 0, st, n, (const float *)u0, status, bd, (float *)held);
    }
    HIP_TRY(s, hipGetLastError());
    return 0;
}

int nmpc_adjoint_sensitivities_device(nmpc_solver *s, int B, const void *x, const void *u, const void *lam, void *out,
                                      int continuous, void *hip_stream)
{
    if (!s) return NMPC_EARG;
    if (B < 1 || !x || !u || !lam || !out) return s->fail(NMPC_EARG, "adjoint_sensitivities: bad arguments");
    if (!continuous && s->cfg.sim_num_steps > ADJ_MAX_STEPS)
        return s->fail(NMPC_EARG, "adjoint_sensitivities: sim_num_steps > %d is not built", ADJ_MAX_STEPS);
    HIP_TRY(s, hipSetDevice(s->cfg.device));
    hipStream_t st = (hipStream_t)hip_stream;
    const dim3 grid((B + 63) / 64), block(64);
    if (s->cfg.dtype == NMPC_DTYPE_F64) {
        Consts<double> c;
        fill_consts(s->cfg, c);
        hipLaunchKernelGGL(k_adjoint_sens<double>, grid, block, 0, st, c, B, (const double *)x, (const double *)u, (const double *)lam, (double *)out, continuous);
    } else {
        Consts<float> c;
        fill_consts(s->cfg, c);
        hipLaunchKernelGGL(k_adjoint_sens<float>, grid, block, 0, st, c, B, (const float *)x, (const float *)u, (const float *)lam, (float *)out, continuous);
    }
    HIP_TRY(s, hipGetLastError());
    return 0;
}

int nmpc_kkt_report_device(nmpc_solver *s, int B, const void *x_traj, const void *u_traj, const void *yref, const void *yref_e,
                           int yref_bcast, void *res, void *hip_stream)
{
    if (!s) return NMPC_EARG;
    if (B < 1 || !x_traj || !u_traj || !yref || !yref_e || !res) return s->fail(NMPC_EARG, "kkt_report: bad arguments");
    if (s->cfg.sim_num_steps > ADJ_MAX_STEPS) return s->fail(NMPC_EARG, "kkt_report: sim_num_steps > %d is not built", ADJ_MAX_STEPS);
    HIP_TRY(s, hipSetDevice(s->cfg.device));
    hipStream_t st = (hipStream_t)hip_stream;
    const dim3 grid((B + 63) / 64), block(64);
    if (s->cfg.dtype == NMPC_DTYPE_F64) {
        Consts<double> c;
        fill_consts(s->cfg, c);
        hipLaunchKernelGGL(k_kkt_report<double>, grid, block, 0, st, c, B, (const double *)x_traj, (const double *)u_traj, (const double *)yref,
                           (const double *)yref_e, yref_bcast, (double *)res);
    } else {
        Consts<float> c;
        fill_consts(s->cfg, c);
        hipLaunchKernelGGL(k_kkt_report<float>, grid, block, 0, st, c, B, (const float *)x_traj, (const float *)u_traj, (const float *)yref,
                           (const float *)yref_e, yref_bcast, (float *)res);
    }
    HIP_TRY(s, hipGetLastError());
    return 0;
}

// Parallel-in-time Riccati factorisation (nmpc_block.hpp) of the LQ problem the LAST solve of this handle ended on: its
// per-stage linearisation (stage tiles in the workspace) and the pin set its last forward sweep left.
int nmpc_block_factor_device(nmpc_solver *s, int B, int blocks, const void *x0, const void *yref, const void *yref_e, int yref_bcast,
                             const void *x_init, const void *u_init, double *factors_out, double *boundary_out, double *check_out,
                             float *ms_out, void *hip_stream)
{
    if (!s) return NMPC_EARG;
    const int N = s->cfg.N;
    if (B < 1 || B > s->cfg.max_batch || !x0 || !yref || !yref_e) return s->fail(NMPC_EARG, "block_factor: bad arguments");
    if (blocks < 1 || blocks > N) return s->fail(NMPC_EARG, "block_factor: blocks=%d outside [1, N=%d]", blocks, N);
    if ((x_init == nullptr) != (u_init == nullptr)) return s->fail(NMPC_EARG, "block_factor: x_init and u_init must both be given or both be NULL");
    if (!(s->cfg.flags & NMPC_FLAG_TEAM_MAPPING) || (s->cfg.flags & NMPC_FLAG_CONDENSED_QP) || !s->tAB)
        return s->fail(NMPC_EARG, "block_factor: FP64 arithmetic in the team mapping only");
    if (!s->solved || s->last_B < B) return s->fail(NMPC_EARG, "block_factor: no solve of >= %d instances on this handle yet", B);
    if (x_init == nullptr && (s->cfg.flags & NMPC_FLAG_SHARE_COLD_START))
        return s->fail(NMPC_EARG, "block_factor: needs the per-stage linearisation (a warm start, or NMPC_FLAG_SHARE_COLD_START off)");
    // ... of the LAST solve: the sweeps read the stage tiles that solve left in the workspace, whatever this call's arguments say
    if (s->last_shared)
        return s->fail(NMPC_EARG, "block_factor: the last solve on this handle shared one cold-start linearisation - only stage-0 tiles exist; "
                                  "solve with a warm start (or NMPC_FLAG_SHARE_COLD_START off) first");
    if (s->cfg.sim_num_steps > AS_MAX_STEPS) return s->fail(NMPC_EARG, "block_factor: sim_num_steps > %d is not built", AS_MAX_STEPS);
    HIP_TRY(s, hipSetDevice(s->cfg.device));
    hipStream_t st = (hipStream_t)hip_stream;
    const int J = blocks, M = (N + J - 1) / J;
    const int Jeff = (N + M - 1) / M;            // blocks that actually hold stages
    const size_t Bw = (size_t)s->Bp + 1;
    if (Jeff > s->blk_J || !s->blk_fac) {
        for (double **p : {&s->blk_agg, &s->blk_bnd, &s->blk_chk, &s->blk_fac})
            if (*p) { (void)gfree(s, *p); *p = nullptr; }
        s->blk_J = 0;
        HIP_TRY(s, gmalloc(s, (void **)&s->blk_agg, Bw * Jeff * 3 * BLK_MAT * sizeof(double)));
        HIP_TRY(s, gmalloc(s, (void **)&s->blk_bnd, Bw * (Jeff + 1) * BLK_MAT * sizeof(double)));
        HIP_TRY(s, gmalloc(s, (void **)&s->blk_chk, Bw * (Jeff + 1) * BLK_MAT * sizeof(double)));
        HIP_TRY(s, gmalloc(s, (void **)&s->blk_fac, Bw * (size_t)N * BLK_FAC_ROWS * sizeof(double)));
        for (auto &e : s->blk_ev)
            if (!e) HIP_TRY(s, hipEventCreate(&e));
        s->blk_J = Jeff;             // (only now: a failed event creation must not leave a "ready" state with null events behind)
    }
    BlockLaunch bl{};
    bl.cp = (const Consts<double> *)s->d_consts;
    bl.g.tAB = (const double *)s->tAB; bl.g.tIV = (const double *)s->iv;
    bl.g.agg = s->blk_agg; bl.g.bnd = s->blk_bnd; bl.g.bchk = check_out ? s->blk_chk : nullptr; bl.g.fac = s->blk_fac;
    bl.g.Bp = s->Bp; bl.g.B = B; bl.g.J = Jeff; bl.g.M = M;
    bl.stream = st; bl.timing = ms_out != nullptr;
    for (int i = 0; i < 4; i++) bl.ev[i] = s->blk_ev[i];
    if (s->cfg.dtype == NMPC_DTYPE_F32IO) {
        Inputs<float> in;
        in.x0 = (const float *)x0; in.yref = (const float *)yref; in.yref_e = (const float *)yref_e;
        in.x_init = (const float *)x_init; in.u_init = (const float *)u_init; in.yref_bcast = yref_bcast;
        HIP_TRY(s, (hipError_t)(s->block_noflag ? launch_block_factor(bl, in) : launch_block_factor_flag(bl, in)));
    } else {
        Inputs<double> in;
        in.x0 = (const double *)x0; in.yref = (const double *)yref; in.yref_e = (const double *)yref_e;
        in.x_init = (const double *)x_init; in.u_init = (const double *)u_init; in.yref_bcast = yref_bcast;
        HIP_TRY(s, (hipError_t)(s->block_noflag ? launch_block_factor(bl, in) : launch_block_factor_flag(bl, in)));
    }
    if (factors_out) HIP_TRY(s, hipMemcpyAsync(factors_out, s->blk_fac, (size_t)B * N * BLK_FAC_ROWS * sizeof(double), hipMemcpyDeviceToDevice, st));
    // boundary_out / check_out are the CALLER's [B][blocks + 1][256]: rows 0 .. Jeff of every instance are written (Jeff <= blocks is only
    // known after the call), each instance at its own stride
    const size_t row_in = (size_t)(Jeff + 1) * BLK_MAT * sizeof(double), row_out = (size_t)(blocks + 1) * BLK_MAT * sizeof(double);
    if (boundary_out) HIP_TRY(s, hipMemcpy2DAsync(boundary_out, row_out, s->blk_bnd, row_in, row_in, (size_t)B, hipMemcpyDeviceToDevice, st));
    if (check_out) HIP_TRY(s, hipMemcpy2DAsync(check_out, row_out, s->blk_chk, row_in, row_in, (size_t)B, hipMemcpyDeviceToDevice, st));
    if (ms_out) {
        HIP_TRY(s, hipEventSynchronize(s->blk_ev[3]));
        for (int i = 0; i < 3; i++) HIP_TRY(s, hipEventElapsedTime(&ms_out[i], s->blk_ev[i], s->blk_ev[i + 1]));
    }
    return Jeff;
}

// The factors (Mbar' tiles | L^-1 tile: 80 doubles per stage) the last solve's own factor sweeps left in the workspace, to the HOST:
// [B][N][80].  Complete only when the solve ran without the LDS stage cache (NMPC_TEAM_LSTG=0): cached stages never reach HBM.
int nmpc_debug_factors(nmpc_solver *s, int B, double *host_out)
{
    if (!s) return NMPC_EARG;
    if (B < 1 || B > s->Bp || !host_out) return s->fail(NMPC_EARG, "debug_factors: bad arguments");
    if (s->wsz != 8) return s->fail(NMPC_EARG, "debug_factors: FP64 workspace only");
    HIP_TRY(s, hipSetDevice(s->cfg.device));
    HIP_TRY(s, hipDeviceSynchronize());
    HIP_TRY(s, hipMemcpy2D(host_out, BLK_FAC_ROWS * sizeof(double), (const double *)s->LM + TLM_MT, TLM_ROWS * sizeof(double),
                           BLK_FAC_ROWS * sizeof(double), (size_t)B * s->cfg.N, hipMemcpyDeviceToHost));
    return 0;
}

// diagnostic: where the block-parallel tail left the instances of the last solve - per instance 0 not in the work list (or no tail),
// 3 finished by the tail, 5 handed to the fallback list; returns the number of blocks the tail uses (0: this handle has no tail)
int nmpc_debug_tail_states(nmpc_solver *s, int B, int32_t *host_out)
{
    if (!s) return NMPC_EARG;
    if (B < 1 || B > s->Bp || !host_out) return s->fail(NMPC_EARG, "debug_tail_states: bad arguments");
    for (int i = 0; i < B; i++) host_out[i] = 0;
    if (s->tail_J == 0) return 0;
    HIP_TRY(s, hipSetDevice(s->cfg.device));
    HIP_TRY(s, hipDeviceSynchronize());
    std::vector<double> ts((size_t)B * TS_ROWS);
    HIP_TRY(s, hipMemcpy(ts.data(), s->d_ts, ts.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < B; i++) host_out[i] = (int32_t)ts[(size_t)i * TS_ROWS];
    return s->tail_J;
}

int nmpc_plant_step_device(nmpc_solver *s, int B, const void *x, const void *u, void *x_next, int normalize_q,
                           void *hip_stream)
{
    if (!s) return NMPC_EARG;
    if (B < 1 || !x || !u || !x_next) return s->fail(NMPC_EARG, "plant_step: bad arguments");
    HIP_TRY(s, hipSetDevice(s->cfg.device));
    hipStream_t st = (hipStream_t)hip_stream;
    const dim3 grid((B + 63) / 64), block(64);
    if (s->cfg.dtype == NMPC_DTYPE_F64) {
        Consts<double> c;
        fill_consts(s->cfg, c);
        hipLaunchKernelGGL(k_plant_step<double>, grid, block, 0, st, c, B, (const double *)x, (const double *)u, (double *)x_next, normalize_q);
    } else {
        Consts<float> c;
        fill_consts(s->cfg, c);
        hipLaunchKernelGGL(k_plant_step<float>, grid, block, 0, st, c, B, (const float *)x, (const float *)u, (float *)x_next, normalize_q);
    }
    HIP_TRY(s, hipGetLastError());
    return 0;
}

int nmpc_hold_and_step_device(nmpc_solver *s, int B, const void *u0, const int32_t *status, void *held, void *x,
                              int normalize_q, void *hip_stream)
{
    if (!s) return NMPC_EARG;
    if (B < 1 || !u0 || !status || !held || !x) return s->fail(NMPC_EARG, "hold_and_step: bad arguments");
    HIP_TRY(s, hipSetDevice(s->cfg.device));
    hipStream_t st = (hipStream_t)hip_stream;
    const dim3 grid((B + 63) / 64), block(64);
    if (s->cfg.dtype == NMPC_DTYPE_F64) {
        Consts<double> c;
        fill_consts(s->cfg, c);
        Bounds4<double> bd;
        for (int i = 0; i < 4; i++) { bd.lb[i] = s->cfg.lbu[i]; bd.ub[i] = s->cfg.ubu[i]; }
        hipLaunchKernelGGL(k_hold_and_step<double>, grid, block, 0, st, c, B, (const double *)u0, status, bd, (double *)held,
                           (double *)x, normalize_q);
    } else {
        Consts<float> c;
        fill_consts(s->cfg, c);
        Bounds4<float> bd;
        for (int i = 0; i < 4; i++) { bd.lb[i] = (float)s->cfg.lbu[i]; bd.ub[i] = (float)s->cfg.ubu[i]; }
        hipLaunchKernelGGL(k_hold_and_step<float>, grid, block, 0, st, c, B, (const float *)u0, status, bd, (float *)held,
                           (float *)x, normalize_q);
    }
    HIP_TRY(s, hipGetLastError());
    return 0;
}

const int32_t *nmpc_device_iterations(nmpc_solver *s) { return s ? s->d_iters : nullptr; }
const int32_t *nmpc_device_passes(nmpc_solver *s) { return s ? s->d_npol : nullptr; }

// host copies of the two arrays above, through THIS library's HIP runtime: [n] int32 each, either may be NULL
int nmpc_get_counts(nmpc_solver *s, int n, int32_t *iterations, int32_t *passes)
{
    if (!s) return NMPC_EARG;
    if (n < 0 || n > s->Bp) return s->fail(NMPC_EARG, "get_counts: n=%d outside [0, %d]", n, s->Bp);
    HIP_TRY(s, hipSetDevice(s->cfg.device));
    HIP_TRY(s, hipDeviceSynchronize());
    if (iterations && n) HIP_TRY(s, hipMemcpy(iterations, s->d_iters, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (passes && n) HIP_TRY(s, hipMemcpy(passes, s->d_npol, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return 0;
}

#ifdef NMPC_PROFILE
// diagnostic builds only (tools/profile_sweeps.py): int64 [8][Bp] DEVICE pointer and its row stride
const long long *nmpc_debug_prof(nmpc_solver *s, int *stride) { if (stride) *stride = s->Bp; return s->d_prof; }
// copies the [8][Bp] stamp table to the host through THIS library's HIP runtime (a second runtime loaded
// by the caller must not touch the pointer); returns the row stride Bp, or a negative error code
int nmpc_debug_prof_copy(nmpc_solver *s, long long *host)
{
    if (!s || !host || !s->d_prof) return NMPC_EARG;
    HIP_TRY(s, hipSetDevice(s->cfg.device));
    HIP_TRY(s, hipDeviceSynchronize());
    HIP_TRY(s, hipMemcpy(host, s->d_prof, (size_t)8 * s->Bp * sizeof(long long), hipMemcpyDeviceToHost));
    return s->Bp;
}
#endif

int nmpc_set_timing(nmpc_solver *s, int on)
{
    if (!s) return NMPC_EARG;
    s->timing = on != 0;
    return 0;
}

int nmpc_get_stats(nmpc_solver *s, nmpc_stats *out)
{
    if (!s || !out) return NMPC_EARG;
    std::memset(out, 0, sizeof(*out));
    out->workspace_bytes = s->ws_bytes;
    if (!s->solved) return 0;
    HIP_TRY(s, hipSetDevice(s->cfg.device));
    HIP_TRY(s, hipDeviceSynchronize());
    if (s->timed) {
        float a = 0, b = 0;
        if (s->timed_fused) {        // one kernel: preparation and QP phase are not separable
            HIP_TRY(s, hipEventElapsedTime(&b, s->ev[0], s->ev[2]));
        } else {
            HIP_TRY(s, hipEventElapsedTime(&a, s->ev[0], s->ev[1]));
            HIP_TRY(s, hipEventElapsedTime(&b, s->ev[1], s->ev[2]));
        }
        out->ms_prepare = a;
        out->ms_solve = b;
        if (s->timed_split) {
            float t = 0;
            HIP_TRY(s, hipEventElapsedTime(&t, s->ev[2], s->ev[3]));
            out->ms_tail = t;
        }
    }
    const int B = s->last_B;
    std::vector<int32_t> it(B), st(B);
    HIP_TRY(s, hipMemcpy(it.data(), s->d_iters, (size_t)B * 4, hipMemcpyDeviceToHost));
    HIP_TRY(s, hipMemcpy(st.data(), s->d_status, (size_t)B * 4, hipMemcpyDeviceToHost));
    out->batch = B;
    out->iter_min = *std::min_element(it.begin(), it.end());
    out->iter_max = *std::max_element(it.begin(), it.end());
    double sum = 0;
    for (int i = 0; i < B; i++) {
        sum += it[i];
        if (st[i] >= 0 && st[i] < 5) out->n_status[st[i]]++;
    }
    out->iter_mean = sum / B;
    std::vector<int32_t> np(B);
    HIP_TRY(s, hipMemcpy(np.data(), s->d_npol, (size_t)B * 4, hipMemcpyDeviceToHost));
    double psum = 0;
    for (int i = 0; i < B; i++) {
        const int a = std::abs(np[i]);
        psum += a;
        out->polish_max = std::max(out->polish_max, a);
        out->n_polished += (np[i] > 0);
    }
    out->polish_mean = psum / B;
    // handed to the general kernel = not settled by the first active-set attempt: such an instance goes on with at least one
    // interior-point iteration (the active-set kernel itself makes none)
    if (s->last_split)
        for (int i = 0; i < B; i++) out->n_tail += it[i] > 0;
    return 0;
}

}  // extern "C"
