// nmpc_team.hpp -- QP phase with 16 lanes per MPC instance ("team" mapping), gfx950 device code.
//
// Why: with one instance per lane a batch of 4096 is only 64 waves on a chip with 1024 SIMDs and
// each wave streams ~40 KB of private workspace per instance through HBM/L2 -- measured
// memory-latency bound (profiles/).  Here a wave holds 4 instances ("teams"); team b owns
// lanes {16a + 4b + c}, lane (a,c) being row r = 4a + c of the 13x13 Riccati matrix and of the stage
// matrices in the row-per-lane sweeps, and element (a,c) of every 4x4 register tile in the tile-form
// sweeps (the FP64 factor and forward sweeps run on v_mfma_f64_4x4x4_4b_f64, whose block layout this
// is).  One wave per workgroup makes every LDS exchange a single-wave hand-off.  B = 4096 -> 1024
// waves = one per SIMD.
//
// Same algorithm and constants as lane_ipm() in nmpc_ipm.hpp and as the oracle; only the
// distribution of the arithmetic over lanes differs ([UPSTREAM] HPIPM Riccati IPM, reached by
// the reference through AcadosOcpSolver.solve(), controller.py:447).  Arithmetic short-cuts that
// change results at rounding level only (tests hold 1e-9 against the oracle): slack reciprocals
// (v_rcp_f64 + 2 Newton steps) replace the ~130 IEEE divisions per stage and iteration, the
// Cholesky pivots use v_rsq_f64 + 2 Newton steps, and the tile form sums in MFMA order.
//
// Per-instance scratch in HBM is "array of structures" (a team reads contiguous runs):
//   tLM [inst][stage][160]: M column-major [13][4] (52) | L packed, diagonal inverted (10) | m (4) | xhat (13) | pad |
//                           tile form of the FP64 path: Mbar^T tiles (64) | L^-1 tile (16)
//   tIV [inst][stage][20] : u | lam_l | lam_u | u_aff | du   (4 each); during an active-set pass the last
//                           two hold the candidate inputs and the pin codes (-1 lower, 0 free, +1 upper)
//   tP  [inst][1 + ckpt][256] : (P_k, p_k), k = 1..ckpt, as left by an active-set pass (16 tiles x 16 lanes of
//                           Pbar in the tile form, 13 rows of 14 in the row form).  The factorisation
//                           of stage k depends on the pins of stages >= k only, so the next pass restarts its
//                           backward sweep at the highest stage whose pin set changed - when that lies in the
//                           checkpointed window (saturation sits in the first stages of the horizon) - instead
//                           of at N-1.  The window bounds the store traffic: 1.4 KB per stage and instance
//                           through a 64 B/clk store path cost 7 % of the sweep when every stage was kept.
// The stage matrices come from the per-instance block tAB written by team_prepare.
#pragma once

#include "nmpc_lane.hpp"

namespace nmpc {

constexpr int TEAM = 16;            // lanes per instance
constexpr int TEAMS_PER_WAVE = 4;
constexpr int TLM_ROWS = 240;      // M (52) | L (10) | m (4) | xhat of the polish sweep (13) | pad (1) |
                                   // tile form: Mbar^T as 4 tiles x 16 lanes (64) | L^-1 tile (16) |
                                   // stages with pins: (B'Pbar Abar)^T unmasked (64) | (B'PB)^T (16)
constexpr int TLM_MT = 80, TLM_Z = 144, TLM_G = 160;
constexpr int TLM_RINV = 52;       // tile form with H_uu = L D L': the four 1 / d_a of a stage (52..63: the slot of the row form's L | m, unused there)
// TAB_ROWS (nmpc_lane.hpp): 104 (Ad rows, 8 each) + 52 (B rows) + 13 (b) + pad here; 12 tiles x 16 in the active-set kernel
constexpr int TP_ROW = 14;          // one row of the Riccati matrix P_k (13) and p_k, per stage checkpoint
constexpr int TP_ROWS = 256;        // per stage: 13 x 14 (VALU form) or 16 tiles x 16 lanes (MFMA form)
// LDS carve per team, in elements of T
constexpr int L_AD = 0;             // [16][8]   rows of the dense A columns
constexpr int L_B = L_AD + 128;     // [16][4]
constexpr int L_BV = L_B + 64;      // [16]
constexpr int L_PB = L_BV + 16;     // [16][4]
constexpr int L_H = L_PB + 64;      // [16]
constexpr int L_PA = L_H + 16;      // [16][14]  rows of P*A
constexpr int L_HG = L_PA + 224;    // [16]      Huu (10) | gu (4)
constexpr int L_MC = L_HG + 16;     // [16][4]   columns of M
constexpr int L_D = L_MC + 64;      // [4] D | [4] rhat | [4] free mask | [4] pinned value
constexpr int L_Y = L_D + 16;       // [2][16][4] partial products M[:,c]*x_c, double buffered
constexpr int L_XH = L_Y + 128;     // [2][16]
constexpr int L_DR = L_XH + 32;     // [2][4]
constexpr int L_RED = L_DR + 8;     // [32] small reductions
constexpr int L_Z = L_RED + 32;     // [32] the 28 entries of the (q,omega)x(q,omega) block of A'PA
// 856 elements: as bytes (6848 B FP64 / 3424 B FP32) the team stride is 192 B resp. 96 B past a
// multiple of the 256-B LDS bank row, so the four teams of a wave - whose lanes are interleaved in
// every 16-lane service group and issue the same relative address at the same time - start 0, 192,
// 128, 64 B (FP32: 0, 96, 192, 32 B) into a bank row and never share a bank on a 16-byte access.
// (A stride of 800 doubles = 25 bank rows made every broadcast read a 4-way conflict:
// SQ_LDS_BANK_CONFLICT was 20 % of the wave cycles; 848 = 128 B past a row is 2-way with this lane map.)
constexpr int TEAM_LDS = L_Z + 48;

// Tail state of a long-horizon work-list instance between the launches of the block-parallel tail: one row of
// TS_ROWS doubles per instance
constexpr int TS_ROWS = 16;
enum { TS_NONE = 0, TS_IPM = 1, TS_AS = 2, TS_DONE = 3, TS_FALLBACK = 4, TS_LISTED = 5 };
// [0] state  [1] passes spent  [2] passes of the attempt in flight  [3] growth reference  [4] mu  [5] rho  [6] interior-point iterations
// [7] threshold of the next attempt  [8] last step  [9] the iterate descends from a warm start
// [10] the block aggregates in memory belong to the pin set of the last pass (blocks whose pins did not change keep theirs)

// record a block's forward sweep leaves for the decision of the pass (block-parallel forward sweep of the tail): FR_ROWS doubles per
// (instance, block): [0] a pin code changed  [1] NaN seen  [2] highest stage with a changed pin code  [3] a pinned input met
// [4..7] candidate command (block 0)  [8..23] state at the end of the block, natural rows (the last block's is xhat_N)
constexpr int FR_ROWS = 24;

template <class T>
struct TeamWork {
    T *tLM;
    T *tIV;
    T *tP;     // [inst][1 + ckpt][13][14] Riccati checkpoints of the active-set passes, or null
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

// Stage vectors of the team mapping are "array of structures": [inst][rows] with rows = (N+1)*13 (xl),
// N*4 (ul), N*17+13 (qr) - a team touches contiguous 13- and 4-element runs (the lane kernels keep
// the same buffers as [row][Bp], which is what coalesces for one instance per lane).
#define NMPC_TLD(p, rows, row) ((p)[(size_t)inst * (size_t)(rows) + (size_t)(row)])
#define NMPC_TST(p, rows, row, v) ((p)[(size_t)inst * (size_t)(rows) + (size_t)(row)] = (v))
// linearisation input u_k: identically zero under the shared cold start (nothing is staged for it then)
#define NMPC_UL0(row) (SHARED ? T(0) : NMPC_TLD(w.ul, ULR, row))

// hard fence for the machine scheduler: nothing is moved across it (used to keep LDS reads batched)
#define NMPC_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ float fast_rcp(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ double fast_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    y = __builtin_fma(0.5 * y, __builtin_fma(-x * y, y, 1.0), y);
    y = __builtin_fma(0.5 * y, __builtin_fma(-x * y, y, 1.0), y);
    return y;
}
__device__ __forceinline__ float fast_rsqrt(float x)
{
    float y = __builtin_amdgcn_rsqf(x);
    return __builtin_fmaf(0.5f * y, __builtin_fmaf(-x * y, y, 1.0f), y);
}

// D = A^T * B + C on 4x4 tiles of four independent blocks (= the four teams of a wave): element (a,c) of
// every operand and of the result sits in lane 16a + 4b + c of block b, so the A operand is read as the
// TRANSPOSE of the tile stored that way (layout and rate probed with tools/probe_mfma: one wave alone
// reaches the full FP64 rate with this instruction, but only half of it with v_fma_f64).
__device__ __forceinline__ double mfma44(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float mfma44(float, float, float c) { return c; }   // FP32 keeps the VALU form
// (-X)'Y + C: the FP64 MFMAs take a negation per operand in their BLGP field (neg:[1,0,0]) - exact, and one v_xor_b32 per tile saved
__device__ __forceinline__ double mfma44_na(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 1); }
__device__ __forceinline__ float mfma44_na(float, float, float c) { return c; }

// padded state order of the tile form: p(3) _ | v(3) _ | q(4) | omega(3) 1   (index 15 is the homogeneous
// coordinate that carries b, p and the gradients); natural index of element e of tile t, -1 for a pad
__device__ __forceinline__ constexpr int nat_of(int t, int e)
{
    return t == 0 ? (e < 3 ? e : -1) : (t == 1 ? (e < 3 ? 3 + e : -1) : (t == 2 ? 6 + e : (e < 3 ? 10 + e : -1)));
}

template <class T>
__device__ __forceinline__ T pick13(const T *v, int i)
{
    T x = 0;
    NMPC_UNROLL for (int l = 0; l < NX; l++) x = (l == i) ? v[l] : x;
    return x;
}

template <class T>
__device__ __forceinline__ T quad_sum(T x)     // over the 4 lanes 4q..4q+3 (the columns of one tile row)
{
    x += __shfl_xor(x, 1);
    x += __shfl_xor(x, 2);
    return x;
}

template <class T>
__device__ __forceinline__ T sel4(const T *v, int j)
{
    return j == 0 ? v[0] : (j == 1 ? v[1] : (j == 2 ? v[2] : v[3]));
}

// index of entry (a,b), a <= b, in the packed upper triangle of a 7x7 symmetric block
__device__ __forceinline__ constexpr int zidx(int a, int b) { return a * 7 - a * (a - 1) / 2 + (b - a); }

// slack reciprocals and affine-direction pieces of one bound pair (lower, upper) of one input
template <class T>
struct Pair {
    T tl, tu, itl, itu, kl, ku;
    __device__ __forceinline__ Pair(T u, T ll, T lu, T lo, T hi)
    {
        tl = u - lo; tu = hi - u;
        itl = fast_rcp(tl); itu = fast_rcp(tu);
        kl = ll * itl; ku = lu * itu;
    }
};

// BATCH: read the LDS operands of the P*[B b A] products in fenced batches (1 wave per SIMD only)
// SHARED: all stages use one (Ad, B, b) (cold start, NMPC_FLAG_SHARE_COLD_START) - compile time so
// that the per-stage reload code and its address arithmetic do not exist in the shared variant
// TO: element type of the caller's output arrays (float for NMPC_DTYPE_F32IO: FP64 arithmetic on FP32 buffers)
template <class T, bool BATCH, bool SHARED, bool MF, class TO = T>
__device__ __forceinline__ void team_ipm(const Consts<T> &c, const Work<T> &w, const Outputs<TO> &out,
                                         const TeamWork<T> &tw, int B, int tpw, T *smem, long long t_entry = 0,
                                         bool lds_prefilled = false, int inst_ov = -2, bool resume = false)
{
    // inst_ov != -2: the instance of this team comes from a work list (k_team_ipm_list; < 0 = idle team)
    // resume: the first active-set attempt of the instance has been made - and given up - by the active-set
    // kernel (nmpc_team_as.hpp): continue exactly where the single-kernel path would be after that attempt
    // lane -> (team, row): team b owns lanes {16a + 4b + c}, row r = 4a + c.  This is the block layout of
    // v_mfma_f64_4x4x4_4b_f64 (operand/result element (a,c) of block b sits in lane 16a + 4b + c, probed
    // with tools/probe_mfma), so a 4x4 tile of a team's matrices is one register across the team.
    const int tid = threadIdx.x, team = (tid >> 2) & 3, r = ((tid >> 4) << 2) | (tid & 3);
    // uniform constants of the hot loops, pinned to vector registers: as scalars they are spilled with
    // their whole 16-dword kernel-argument tuple and re-read lane by lane inside every stage
    T dt_v = c.dt, kkt_v = c.kkt_tol;
    asm volatile("" : "+v"(dt_v), "+v"(kkt_v));
    const int rr = r < NX ? r : NX - 1;   // row used for loads; rows 13..15 shadow row 12 and never store
    const int j = r & 3;                  // input component handled by lanes r < 4 (others shadow)
    const bool rowl = r < NX, cmpl = r < NU;
    int inst = inst_ov != -2 ? inst_ov : blockIdx.x * tpw + team;   // 1, 2 or 4 live teams per wave (launch decides)
    const bool valid = inst_ov != -2 ? (inst >= 0 && inst < B) : (team < tpw && inst < B);
    if (!valid) inst = B - 1;             // idle teams shadow the last instance and never store
    const int N = c.N;
    const int XLR = (N + 1) * NX, ULR = N * NU, QRR = N * QR_ROWS + NX;
    const int lane = inst;   // index of the profiling slots (NMPC_PROFILE builds)
    (void)lane;
    const T nc = T(2 * NU) * T(N);
    T *S = smem + team * TEAM_LDS;
    T *sAd = S + L_AD, *sB = S + L_B, *sbv = S + L_BV, *sPB = S + L_PB, *sh = S + L_H, *sPA = S + L_PA;
    T *sHg = S + L_HG, *sMc = S + L_MC, *sD = S + L_D, *sY = S + L_Y, *sXh = S + L_XH, *sDr = S + L_DR;
    T *sRed = S + L_RED, *sZ = S + L_Z;
    T *tLM = tw.tLM + (size_t)inst * N * TLM_ROWS, *tIV = tw.tIV + (size_t)inst * N * IV_ROWS;
    const int ckpt = c.polish_ckpt;       // checkpoints exist for stages 1..ckpt
    T *tP = tw.tP ? tw.tP + (size_t)inst * (ckpt + 1) * TP_ROWS : nullptr;
    const int nteams = 4;

    // the two (a,b) entries of the packed 7x7 block this lane computes in the P update
    int za0 = 0, zb0 = 0, za1 = 0, zb1 = 0;
    {
        const int e0 = r < 14 ? 2 * r : 26, e1 = e0 + 1;
        NMPC_UNROLL for (int a = 0; a < NZ; a++) {
            NMPC_UNROLL for (int b = a; b < NZ; b++) {
                if (zidx(a, b) == e0) { za0 = a; zb0 = b; }
                if (zidx(a, b) == e1) { za1 = a; zb1 = b; }
            }
        }
    }
    const int az = rr >= 6 ? rr - 6 : 0;   // column of the packed block this row corresponds to

    // row r of the stage matrices and column r of A, in registers
    T Adrow[NZ], Brow[NU], b_r = 0, Acol[NX];
    // stage data come from the team-friendly copy written by k_prepare:
    // tAB [inst][Ns][TAB_ROWS] = Ad rows [13][8] | B rows [13][4] | b [13]  (a lane reads 64 + 32 + 8 B)
    auto load_stage = [&](int k) {
        const T *a = w.tAB + ((size_t)inst * (SHARED ? 1 : N) + k) * TAB_ROWS;
        NMPC_UNROLL for (int cc = 0; cc < 8; cc++) {
            const T v = a[rr * 8 + cc];
            if (cc < NZ) Adrow[cc] = v;
            sAd[r * 8 + cc] = v;
        }
        NMPC_UNROLL for (int i = 0; i < NU; i++) {
            Brow[i] = a[104 + rr * NU + i];
            sB[r * 4 + i] = Brow[i];
        }
        b_r = a[156 + rr];
        sbv[r] = b_r;
        NMPC_WSYNC();
        NMPC_UNROLL for (int l = 0; l < NX; l++) {
            const T zc = sAd[l * 8 + az];
            const T e = (l == rr) ? T(1) : T(0);
            const T ev = (l == rr - 3) ? dt_v : T(0);
            Acol[l] = rr >= 6 ? zc : (rr >= 3 ? e + ev : e);
        }
    };
    // per-stage linearisation in the tile sweeps: only the LDS copy is needed there, and its global loads
    // are issued one stage ahead
    T pfs[13];
    auto fetch_stage = [&](int k) {
        const T *a = w.tAB + ((size_t)inst * N + k) * TAB_ROWS;
        NMPC_UNROLL for (int cc = 0; cc < 8; cc++) pfs[cc] = a[rr * 8 + cc];
        NMPC_UNROLL for (int i = 0; i < NU; i++) pfs[8 + i] = a[104 + rr * NU + i];
        pfs[12] = a[156 + rr];
    };
    auto put_stage = [&]() {
        NMPC_UNROLL for (int cc = 0; cc < 8; cc++) sAd[r * 8 + cc] = pfs[cc];
        NMPC_UNROLL for (int i = 0; i < NU; i++) sB[r * 4 + i] = pfs[8 + i];
        sbv[r] = pfs[12];
        NMPC_WSYNC();
    };
    // Shared linearisation: the LDS copy stays valid for the whole kernel, so the row / column registers
    // are re-read where a VALU-form sweep needs them instead of living across the (register-hungry)
    // tile-form sweeps.
    auto rows_from_lds = [&]() {
        NMPC_UNROLL for (int cc = 0; cc < NZ; cc++) Adrow[cc] = sAd[rr * 8 + cc];
        NMPC_UNROLL for (int i = 0; i < NU; i++) Brow[i] = sB[rr * 4 + i];
        b_r = sbv[rr];
        NMPC_UNROLL for (int l = 0; l < NX; l++) {
            const T zc = sAd[l * 8 + az];
            const T e = (l == rr) ? T(1) : T(0);
            const T ev = (l == rr - 3) ? dt_v : T(0);
            Acol[l] = rr >= 6 ? zc : (rr >= 3 ? e + ev : e);
        }
    };

    if (SHARED) {
        if (lds_prefilled) { if (!MF) rows_from_lds(); }   // the fused preparation left the stage matrices in LDS
                                                          // (the tile path re-reads its rows where it needs them)
        else load_stage(0);
    }

    const T lbj = sel4(c.lbu, j), ubj = sel4(c.ubu, j), Rdj = sel4(c.Rd, j);
    // ---- active-set guess of the first pass: everything free.  The interior point iterate itself is
    // only written when a team first enters that mode (init_point below): with the polish on, most
    // instances never do.
    if (cmpl && valid) {
        for (int k = 0; k < N; k++) tIV[k * IV_ROWS + 16 + j] = 0;
    }
    // the linearisation inputs are fetched a chunk of stages at a time: the stores may alias them
    auto init_point = [&](bool mine) {
        constexpr int CHI = 10;
        for (int k0 = 0; k0 < N; k0 += CHI) {
            T ulv[CHI];
            NMPC_UNROLL for (int i = 0; i < CHI; i++) ulv[i] = NMPC_UL0( ((k0 + i < N) ? k0 + i : N - 1) * NU + j);
            NMPC_UNROLL for (int i = 0; i < CHI; i++) {
                const int k = k0 + i;
                const T lo = lbj - ulv[i], hi = ubj - ulv[i];
                T thr = c.thr0;
                if (c.thr0_rel * (hi - lo) > thr) thr = c.thr0_rel * (hi - lo);
                if (hi - lo < T(2) * thr) thr = T(0.5) * (hi - lo);
                T v = 0;
                if (v - lo < thr) v = lo + thr;
                if (hi - v < thr) v = hi - thr;
                if (k < N && cmpl && valid && mine) {
                    T *ivk = tIV + k * IV_ROWS;
                    ivk[j] = v;
                    ivk[4 + j] = c.mu0 / (v - lo);
                    ivk[8 + j] = c.mu0 / (hi - v);
                }
            }
        }
    };
    bool have_point = false;
    __syncthreads();
    NMPC_PROF_BEGIN
    NMPC_PROF_SINCE(t_entry)
    T mu = c.mu0, rho = T(1), pol_mu = c.polish_mu;
    int it = 0, status = 0, npol = 0, pass_in_attempt = 0;
    if (resume) {                 // one failed attempt behind us: its passes count, the next one waits for mu <= 1e-2 polish_mu
        npol = w.npol[inst] < 0 ? -w.npol[inst] : w.npol[inst];
        pol_mu *= T(1e-2);
    }
    int k_top = N - 1;      // highest stage this team's next backward sweep has to refactorise
    bool maybe_pins = false; // the current pin set may be non-empty (decides what the forward sweep prefetches)
    int ck_valid = 0;        // checkpoints 1..ck_valid of this team are current
    // per-team mode: interior point iteration, active-set (polish) pass, or finished
    enum { M_IPM = 0, M_POL = 1, M_DONE = 2 };
    int mode = valid ? M_IPM : M_DONE;      // idle teams never hold the wave back
    bool from_ua = false;   // the accepted active-set solution lives in the u_aff slot
    const T Qdr = [&] { T v = 0; NMPC_UNROLL for (int i = 0; i < NX; i++) v = (i == rr) ? c.Qd[i] : v; return v; }();
    const T QdNr = [&] { T v = 0; NMPC_UNROLL for (int i = 0; i < NX; i++) v = (i == rr) ? c.QdN[i] : v; return v; }();

    for (;;) {
        // per-team transitions; the wave keeps sweeping until all its teams are done
        if (mode == M_IPM) {
            if (!(mu == mu)) { status = 1; mode = M_DONE; }
            else if (mu <= c.tol_comp && rho <= c.tol_stat) mode = M_DONE;
            else if (c.polish && mu <= pol_mu && npol < c.polish_budget) { mode = M_POL; pass_in_attempt = 0; k_top = N - 1; maybe_pins = it > 0; }
            else if (it >= c.iter_max) { status = 2; mode = M_DONE; }
        }
        if (__ballot(mode != M_DONE) == 0) break;
        const bool pol = mode == M_POL, ipm = mode == M_IPM;
        const bool act = mode != M_DONE;  // frozen teams keep computing but never store
        const bool st_ok = act && valid;
        if (ipm) it++;
        if (__ballot(ipm && !have_point) != 0) {      // first interior point iteration of some team
            init_point(ipm && !have_point);
            __syncthreads();
        }
        have_point |= ipm;
        // the wave sweeps from the highest stage any of its live teams needs (wave-uniform trip count)
        int ks = N - 1;
        if (tP) {
            if (r == 0) { sRed[28] = (T)(act ? (ipm ? N - 1 : k_top) : -1); sRed[29] = (T)(act ? ck_valid : N); }
            __syncthreads();
            ks = 0;
            int vmin = N;            // every live team must own the checkpoint the wave resumes from
            for (int t = 0; t < nteams; t++) {
                const int kt = (int)smem[t * TEAM_LDS + L_RED + 28], vt = (int)smem[t * TEAM_LDS + L_RED + 29];
                ks = kt > ks ? kt : ks;
                vmin = vt < vmin ? vt : vmin;
            }
            if (ks >= vmin) ks = N - 1;
            ks = __builtin_amdgcn_readfirstlane(ks);      // the same in every lane: make the sweep a scalar loop
        }

        // Checkpoint window of this pass: the first pass of an attempt keeps only the first two stages (almost
        // every instance is done after it, and those that are not usually saturate in stage 0 or 1: writing
        // 12 stages of tiles in every first pass cost 20 % of the throughput at B = 65536); corrected passes
        // keep the configured window.  ck_valid tracks which checkpoints are current: stages this pass
        // recomputes but does not store lose theirs.
        const int wnd = pass_in_attempt == 0 ? (ckpt < 2 ? ckpt : 2) : ckpt;
        if (pol) ck_valid = (wnd < ks) ? wnd : ck_valid;

        // ================= sweep A: backward factorisation, affine right-hand side
        bool ok = true, nanp = false;
        // a wave whose live teams are all in active-set passes skips the barrier terms (wave-uniform)
        const bool any_ipm = __ballot(ipm) != 0;
        T pv = 0, n_ul = 0, n_pc = 0;       // also scratch of the later sweeps
        if constexpr (!MF) {
        T Prow[NX];
        if (ks == N - 1) {
            NMPC_UNROLL for (int cc = 0; cc < NX; cc++) Prow[cc] = (cc == rr) ? QdNr : T(0);
            pv = NMPC_TLD(w.qr, QRR, N * QR_ROWS + rr);
        } else {                      // resume from the checkpoint an earlier active-set pass left
            const T *cp = tP + ((size_t)(ks + 1) * NX + rr) * TP_ROW;
            NMPC_UNROLL for (int cc = 0; cc < NX; cc++) Prow[cc] = cp[cc];
            pv = cp[NX];
        }
        // software prefetch of the next stage's scalars (global loads stay in flight over the stage)
        n_ul = NMPC_UL0( ks * NU + j);
        n_pc = tIV[ks * IV_ROWS + 16 + j];
        T n_u = 0, n_ll = 0, n_lu = 0,
          n_rk = NMPC_TLD(w.qr, QRR, ks * QR_ROWS + NX + j), n_qr = NMPC_TLD(w.qr, QRR, ks * QR_ROWS + rr);
        if (any_ipm) { n_u = tIV[ks * IV_ROWS + j]; n_ll = tIV[ks * IV_ROWS + 4 + j]; n_lu = tIV[ks * IV_ROWS + 8 + j]; }
        for (int k = ks; k >= 0; k--) {
            if (!SHARED) load_stage(k);
            T *lmk = tLM + k * TLM_ROWS;
            const T ul = n_ul, u = n_u, ll = n_ll, lu = n_lu, rk = n_rk, q_r = n_qr, pc = n_pc;
            if (k > 0) {
                const T *ivn = tIV + (k - 1) * IV_ROWS;
                n_ul = NMPC_UL0( (k - 1) * NU + j); n_pc = ivn[16 + j];
                if (any_ipm) { n_u = ivn[j]; n_ll = ivn[4 + j]; n_lu = ivn[8 + j]; }
                n_rk = NMPC_TLD(w.qr, QRR, (k - 1) * QR_ROWS + NX + j); n_qr = NMPC_TLD(w.qr, QRR, (k - 1) * QR_ROWS + rr);
            }
            {   // IPM: barrier terms.  Active-set pass: no barrier, pinned inputs are taken out of B
                // (free mask) and enter through b (pinned value); their own row keeps R_jj so u_j = bound.
                const T lo = lbj - ul, hi = ubj - ul;
                T sg = 0;
                if (any_ipm) {
                    const Pair<T> pr(u, ll, lu, lo, hi);
                    sg = pol ? T(0) : pr.kl + pr.ku;
                }
                const bool pinned = pol && pc != T(0);
                const T vpin = pc < T(0) ? lo : hi;
                if (cmpl) {
                    sD[j] = Rdj + sg;
                    sD[4 + j] = pinned ? -Rdj * vpin : (pol ? rk : rk - sg * u);
                    sD[8 + j] = pinned ? T(0) : T(1);
                    sD[12 + j] = pinned ? vpin : T(0);
                }
            }
            // P1: row r of P*B, P*b + p, P*A.  The LDS operands are read in three batches with the
            // scheduler fenced in between: left alone, hipcc interleaves "one ds_read, wait, two
            // FMAs" and exposes the full LDS latency on every read (one wave per SIMD: nothing
            // else to run).  Batched, 30-odd reads are in flight and the counted lgkmcnt waits
            // consume them in order.
            T PBrow[NU], h = pv, PArow[NX];
            NMPC_UNROLL for (int i = 0; i < NU; i++) PBrow[i] = 0;
            NMPC_UNROLL for (int cc = 0; cc < NZ; cc++) PArow[6 + cc] = 0;
            if (!BATCH) {   // several waves per SIMD hide the latency; keep the register footprint small
                NMPC_UNROLL for (int l = 0; l < NX; l++) {
                    const T pl = Prow[l];
                    NMPC_UNROLL for (int i = 0; i < NU; i++) PBrow[i] += pl * sB[l * 4 + i];
                    h += pl * sbv[l];
                    NMPC_UNROLL for (int cc = 0; cc < NZ; cc++) {
                        if (l < ad_rows(cc)) PArow[6 + cc] += pl * sAd[l * 8 + cc];
                    }
                }
            } else {
                T Bl[NX][NU], bl[NX];
                NMPC_UNROLL for (int l = 0; l < NX; l++) {
                    NMPC_UNROLL for (int i = 0; i < NU; i++) Bl[l][i] = sB[l * 4 + i];
                    bl[l] = sbv[l];
                }
                NMPC_SCHED_FENCE();
                NMPC_UNROLL for (int l = 0; l < NX; l++) {
                    NMPC_UNROLL for (int i = 0; i < NU; i++) PBrow[i] += Prow[l] * Bl[l][i];
                    h += Prow[l] * bl[l];
                }
            }
            if (BATCH) NMPC_UNROLL for (int half = 0; half < 2; half++) {
                const int l0 = half ? 7 : 0, l1 = half ? NX : 7;
                T Al[7][8];
                NMPC_SCHED_FENCE();
                NMPC_UNROLL for (int l = l0; l < l1; l++) {
                    NMPC_UNROLL for (int cc = 0; cc < 8; cc++) Al[l - l0][cc] = sAd[l * 8 + cc];
                }
                NMPC_SCHED_FENCE();
                NMPC_UNROLL for (int l = l0; l < l1; l++) {
                    NMPC_UNROLL for (int cc = 0; cc < NZ; cc++) {
                        if (l < ad_rows(cc)) PArow[6 + cc] += Prow[l] * Al[l - l0][cc];
                    }
                }
            }
            NMPC_UNROLL for (int i = 0; i < 3; i++) { PArow[i] = Prow[i]; PArow[3 + i] = dt_v * Prow[i] + Prow[3 + i]; }
            NMPC_UNROLL for (int i = 0; i < NU; i++) { h += PBrow[i] * sD[12 + i]; PBrow[i] *= sD[8 + i]; }
            NMPC_UNROLL for (int i = 0; i < NU; i++) sPB[r * 4 + i] = PBrow[i];
            sh[r] = h;
            NMPC_UNROLL for (int cc = 0; cc < NX; cc++) sPA[r * 14 + cc] = PArow[cc];
            NMPC_WSYNC();
            // P2: lanes 0..9 one entry of Huu = D + B'PB each, lanes 10..13 one entry of gu = rhat + B'h;
            // plus the two (z,z) dot products of A'(PA) owned by this lane.  All LDS operands first.
            {
                const int e = r < 14 ? r : 13;
                const int ei = e < 10 ? (e >= 6 ? 3 : (e >= 3 ? 2 : (e >= 1 ? 1 : 0))) : e - 10;
                const int ej = e < 10 ? e - ei * (ei + 1) / 2 : 0;
                const T *rhs = e < 10 ? sPB + ej : sh;
                const int rst = e < 10 ? 4 : 1;
                T xa[NX], xb[NX], a0[NX], t0[NX], a1[NX], t1[NX];
                if (BATCH) NMPC_SCHED_FENCE();
                NMPC_UNROLL for (int l = 0; l < NX; l++) { xa[l] = sB[l * 4 + ei]; xb[l] = rhs[l * rst]; }
                if (k > 0) {
                    NMPC_UNROLL for (int l = 0; l < NX; l++) {
                        a0[l] = sAd[l * 8 + za0]; t0[l] = sPA[l * 14 + 6 + zb0];
                        a1[l] = sAd[l * 8 + za1]; t1[l] = sPA[l * 14 + 6 + zb1];
                    }
                }
                const T a_c = e < 10 ? (ei == ej ? sD[ei] : T(0)) : sD[4 + ei], mk = sD[8 + ei];
                if (BATCH) NMPC_SCHED_FENCE();
                T a = 0;
                NMPC_UNROLL for (int l = 0; l < NX; l++) a += xa[l] * xb[l];
                sHg[r] = a_c + mk * a;
                if (k > 0) {
                    T d0 = 0, d1 = 0;
                    NMPC_UNROLL for (int l = 0; l < NX; l++) { d0 += a0[l] * t0[l]; d1 += a1[l] * t1[l]; }
                    if (r < 14) { sZ[2 * r] = d0; sZ[2 * r + 1] = d1; }
                }
            }
            NMPC_WSYNC();
            // P3: Cholesky (replicated), column r of M, p_k
            T Lf[10], mv[NU], Mcol[NU];
            NMPC_UNROLL for (int i = 0; i < 10; i++) Lf[i] = sHg[i];
            NMPC_UNROLL for (int i = 0; i < NU; i++) mv[i] = sHg[10 + i];
            NMPC_UNROLL for (int jj = 0; jj < NU; jj++) {
                T d = Lf[lidx(jj, jj)];
                NMPC_UNROLL for (int l = 0; l < jj; l++) d -= Lf[lidx(jj, l)] * Lf[lidx(jj, l)];
                const bool pos = d > T(0);     // branch-free: keeps the stage body one basic block
                ok &= pos; nanp |= !(d == d); d = pos ? d : T(1);
                const T rd = fast_rsqrt(d);
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
            // P4: row r of P_k = Q + A'(PA) - M'M.  A = [I dt*I X; 0 I Y; 0 0 Z]: position rows of
            // A'(PA) are rows of PA, velocity rows add dt times the position row, and the (q,omega)
            // rows are the transposed (p,v)x(q,omega) entries plus the 28 dot products in sZ.
            if (k > 0) {
                T Pn[NX];
                if (rr < 3) {
                    NMPC_UNROLL for (int cc = 0; cc < NX; cc++) Pn[cc] = PArow[cc];
                } else if (rr < 6) {
                    NMPC_UNROLL for (int cc = 0; cc < NX; cc++) Pn[cc] = PArow[cc] + dt_v * sPA[(rr - 3) * 14 + cc];
                } else {
                    NMPC_UNROLL for (int cc = 0; cc < 3; cc++) {
                        const T tp = sPA[cc * 14 + 6 + az];
                        Pn[cc] = tp;
                        Pn[3 + cc] = sPA[(3 + cc) * 14 + 6 + az] + dt_v * tp;
                    }
                    NMPC_UNROLL for (int cc = 0; cc < NZ; cc++) {
                        const int lo_ = az < cc ? az : cc, hi_ = az < cc ? cc : az;
                        Pn[6 + cc] = sZ[lo_ * 7 - lo_ * (lo_ - 1) / 2 + (hi_ - lo_)];
                    }
                }
                NMPC_UNROLL for (int cc = 0; cc < NX; cc++) {
                    T a = Pn[cc] + ((cc == rr) ? Qdr : T(0));
                    NMPC_UNROLL for (int i = 0; i < NU; i++) a -= Mcol[i] * sMc[cc * 4 + i];
                    Prow[cc] = a;
                }
                pv = pvn;
                if (tP && k <= wnd && pol && st_ok && rowl) {
                    T *cp = tP + ((size_t)k * NX + rr) * TP_ROW;
                    NMPC_UNROLL for (int cc = 0; cc < NX; cc++) cp[cc] = Prow[cc];
                    cp[NX] = pv;
                }
            }
            NMPC_WSYNC();
        }
        } else {
            // ---- tile form (FP64): the stage is ~120 v_mfma_f64_4x4x4 on register tiles and one LDS
            // exchange.  Pbar is the 16x16 matrix [[P p],[p' *]] in the padded order; Abar = [[A b],[0 1]]
            // has dense columns only in the q / omega tiles (Aq0, Aq1; b sits in the pad column of Aq1),
            // so P*A, A'(PA), B'PA, B'PB, the gradients and p_k all come out of the same tile products.
            const int ta = r >> 2, tc = r & 3;
            int natR[4], natC[4];
            NMPC_UNROLL for (int t = 0; t < 4; t++) { natR[t] = nat_of(t, ta); natC[t] = nat_of(t, tc); }
            T Qdg[4];
            NMPC_UNROLL for (int t = 0; t < 4; t++) Qdg[t] = (ta == tc && natR[t] >= 0) ? pick13(c.Qd, natR[t]) : T(0);
            T Aq0[4], Aq1b[4], Bt[4];
            auto load_tiles = [&]() {
                NMPC_UNROLL for (int kt = 0; kt < 4; kt++) {
                    const int l = natR[kt] >= 0 ? natR[kt] : 0;
                    const bool real = natR[kt] >= 0;
                    const T a0 = sAd[l * 8 + tc], a1 = sAd[l * 8 + 4 + (tc < 3 ? tc : 0)], bb = sB[l * 4 + tc], bv_ = sbv[l];
                    Aq0[kt] = real ? a0 : T(0);
                    Aq1b[kt] = real ? (tc < 3 ? a1 : bv_) : ((kt == 3 && ta == 3 && tc == 3) ? T(1) : T(0));
                    Bt[kt] = real ? bb : T(0);
                }
            };
            if (SHARED) load_tiles(); else fetch_stage(ks);
            T Pt[4][4];
            if (ks == N - 1) {
                NMPC_UNROLL for (int it = 0; it < 4; it++) {
                    NMPC_UNROLL for (int jt = 0; jt < 4; jt++) {
                        T v = 0;
                        if (it == jt && ta == tc && natR[it] >= 0) v = pick13(c.QdN, natR[it]);
                        if (jt == 3 && tc == 3 && natR[it] >= 0) v = NMPC_TLD(w.qr, QRR, N * QR_ROWS + natR[it]);
                        if (it == 3 && ta == 3 && natC[jt] >= 0) v = NMPC_TLD(w.qr, QRR, N * QR_ROWS + natC[jt]);
                        Pt[it][jt] = v;
                    }
                }
            } else {                  // resume from the checkpoint an earlier active-set pass left
                const T *cp = tP + (size_t)(ks + 1) * TP_ROWS + r;
                NMPC_UNROLL for (int it = 0; it < 4; it++) {
                    NMPC_UNROLL for (int jt = 0; jt < 4; jt++) Pt[it][jt] = cp[(it * 4 + jt) * 16];
                }
            }
            n_ul = NMPC_UL0( ks * NU + j);
            n_pc = tIV[ks * IV_ROWS + 16 + j];
            T n_u = 0, n_ll = 0, n_lu = 0,
              n_rk = NMPC_TLD(w.qr, QRR, ks * QR_ROWS + NX + j), n_qr = NMPC_TLD(w.qr, QRR, ks * QR_ROWS + rr);
            if (any_ipm) { n_u = tIV[ks * IV_ROWS + j]; n_ll = tIV[ks * IV_ROWS + 4 + j]; n_lu = tIV[ks * IV_ROWS + 8 + j]; }
            for (int k = ks; k >= 0; k--) {
                if (!SHARED) { put_stage(); if (k > 0) fetch_stage(k - 1); load_tiles(); }
                T *lmk = tLM + k * TLM_ROWS;
                const T ul = n_ul, u = n_u, ll = n_ll, lu = n_lu, rk = n_rk, q_r = n_qr, pc = n_pc;
                if (k > 0) {
                    const T *ivn = tIV + (k - 1) * IV_ROWS;
                    n_ul = NMPC_UL0( (k - 1) * NU + j); n_pc = ivn[16 + j];
                    if (any_ipm) { n_u = ivn[j]; n_ll = ivn[4 + j]; n_lu = ivn[8 + j]; }
                    n_rk = NMPC_TLD(w.qr, QRR, (k - 1) * QR_ROWS + NX + j); n_qr = NMPC_TLD(w.qr, QRR, (k - 1) * QR_ROWS + rr);
                }
                bool pinned;
                {
                    const T lo = lbj - ul, hi = ubj - ul;
                    T sg = 0;
                    if (any_ipm) {
                        const Pair<T> pr(u, ll, lu, lo, hi);
                        sg = pol ? T(0) : pr.kl + pr.ku;
                    }
                    pinned = pol && pc != T(0);
                    const T vpin = pc < T(0) ? lo : hi;
                    if (cmpl) {
                        sD[j] = Rdj + sg;
                        sD[4 + j] = pinned ? -Rdj * vpin : (pol ? rk : rk - sg * u);
                        sD[8 + j] = pinned ? T(0) : T(1);
                        sD[12 + j] = pinned ? vpin : T(0);
                    }
                    sh[r] = q_r;                     // natural row rr of the stage gradient
                }
                NMPC_WSYNC();
                const T mask_a = sD[8 + ta], mask_c = sD[8 + tc], D_a = sD[ta], rhat_a = sD[4 + ta];
                T Aq1[4];
                NMPC_UNROLL for (int kt = 0; kt < 4; kt++) Aq1[kt] = Aq1b[kt];
                const bool any_pins = __ballot(pinned) != 0;
                if (any_pins) {                      // pinned inputs enter through b (column 15 of Abar)
                    const T vp = sD[12 + tc];
                    NMPC_UNROLL for (int kt = 0; kt < 4; kt++) {
                        const T sm = quad_sum(Bt[kt] * vp);
                        if (tc == 3 && natR[kt] >= 0) Aq1[kt] += sm;
                    }
                }
                // W = Pbar * [Aq0 | Aq1 | B]  (Pbar symmetric: its tile (kt,it) read transposed is tile (it,kt))
                T W0[4], W1[4], WB[4];
                NMPC_UNROLL for (int it = 0; it < 4; it++) {
                    T a0 = 0, a1 = 0, aB = 0;
                    NMPC_UNROLL for (int kt = 0; kt < 4; kt++) {
                        if (kt < 3) a0 = mfma44(Pt[kt][it], Aq0[kt], a0);     // tile (3,0) of Abar (d omega+ / d q, homogeneous row) is zero
                        a1 = mfma44(Pt[kt][it], Aq1[kt], a1);
                        aB = mfma44(Pt[kt][it], Bt[kt], aB);
                    }
                    W0[it] = a0; W1[it] = a1; WB[it] = aB;
                }
                // column tiles of Pbar*Abar: the p and v columns of Abar are e_p and dt*e_p + e_v
                T PA[4][4];
                NMPC_UNROLL for (int kt = 0; kt < 4; kt++) {
                    PA[kt][0] = Pt[kt][0];
                    PA[kt][1] = dt_v * Pt[kt][0] + Pt[kt][1];
                    PA[kt][2] = W0[kt];
                    PA[kt][3] = W1[kt];
                }
                // X = B'(Pbar Abar) (column 15: B'h), Hr = B'PB
                T X0raw = 0;
                // (its p / v column tiles are transposes of tiles of Pbar B: X' = Abar' Pbar B, one MFMA with
                // the identity each instead of four)
                const T Idt = (ta == tc) ? T(1) : T(0);
                T X[4], Hr = 0;
                NMPC_UNROLL for (int jt = 0; jt < 4; jt++) {
                    T a = 0;
                    if (jt == 0) a = mfma44(WB[0], Idt, T(0));
                    else if (jt == 1) a = mfma44(WB[1], Idt, T(0)) + dt_v * X0raw;
                    else { NMPC_UNROLL for (int kt = 0; kt < 4; kt++) a = mfma44(Bt[kt], PA[kt][jt], a); }
                    if (jt == 0) X0raw = a;
                    X[jt] = mask_a * a;
                    // gradient rows of the pinned inputs for the multiplier check of the forward sweep:
                    // g = R u + r + B'PB (mask u) + B'(Pbar Abar) xbar, unmasked, stored transposed
                    if (any_pins && st_ok) lmk[TLM_G + jt * 16 + tc * 4 + ta] = a;
                }
                NMPC_UNROLL for (int kt = 0; kt < 4; kt++) Hr = mfma44(Bt[kt], WB[kt], Hr);
                if (any_pins && st_ok) lmk[TLM_G + 64 + tc * 4 + ta] = Hr;
                if (tc == 3) X[3] += rhat_a;                                   // gu = rhat + mask * B'h
                const T Huu = ((ta == tc) ? D_a : T(0)) + mask_a * mask_c * Hr;
                if (tc <= ta) sHg[lidx(ta, tc)] = Huu;
                if (tc == 3) sHg[10 + ta] = X[3];
                NMPC_WSYNC();
                // Cholesky (replicated), m = L^-1 gu
                T Lf[10], mv[NU];
                NMPC_UNROLL for (int i = 0; i < 10; i++) Lf[i] = sHg[i];
                NMPC_UNROLL for (int i = 0; i < NU; i++) mv[i] = sHg[10 + i];
#if defined(NMPC_DEBUG_NAN) && defined(__HIP_DEVICE_COMPILE__) && defined(NMPC_PROFILE)
                {   // diagnostic build (tools/dev/nan_probe.py): first (iteration, stage) at which a non-finite value reaches the factor stage
                    auto nf = [](T v) { return !(v - v == T(0)); };
                    bool fa = false, fp = false, fh = false, fd = nf(D_a) || nf(rhat_a);
                    NMPC_UNROLL for (int t = 0; t < 4; t++) fa |= nf(Aq0[t]) || nf(Aq1[t]) || nf(Bt[t]);
                    NMPC_UNROLL for (int a_ = 0; a_ < 4; a_++) { NMPC_UNROLL for (int b_ = 0; b_ < 4; b_++) fp |= nf(Pt[a_][b_]); }
                    NMPC_UNROLL for (int i = 0; i < 10; i++) fh |= nf(Lf[i]);
                    const unsigned long long tm_ = 0x000F000F000F000Full << (4 * team);
                    const int code = (__ballot(fa) & tm_) ? 1 : ((__ballot(fd) & tm_) ? 4 : ((__ballot(fp) & tm_) ? 2 : ((__ballot(fh) & tm_) ? 3 : 0)));
                    if (prof_acc_[7] >= 0 && code) prof_acc_[7] = -(1000000 + it * 10000 + k * 10 + code);
                }
#endif
                NMPC_UNROLL for (int jj = 0; jj < NU; jj++) {
                    T d = Lf[lidx(jj, jj)];
                    NMPC_UNROLL for (int l = 0; l < jj; l++) d -= Lf[lidx(jj, l)] * Lf[lidx(jj, l)];
                    const bool pos = d > T(0);
                    ok &= pos; nanp |= !(d == d); d = pos ? d : T(1);
                    const T rd = fast_rsqrt(d);
                    Lf[lidx(jj, jj)] = rd;
                    NMPC_UNROLL for (int i = jj + 1; i < NU; i++) {
                        T a = Lf[lidx(i, jj)];
                        NMPC_UNROLL for (int l = 0; l < jj; l++) a -= Lf[lidx(i, l)] * Lf[lidx(jj, l)];
                        Lf[lidx(i, jj)] = a * rd;
                    }
                }
                l_solve(Lf, mv);
                // Y = L^-T as a tile (lane (a,c) holds (L^-1 e_a)_c), M = L^-1 X = Y' X
                T ea[NU];
                NMPC_UNROLL for (int i = 0; i < NU; i++) ea[i] = (i == ta) ? T(1) : T(0);
                l_solve(Lf, ea);
                const T Y = sel4(ea, tc);
                T M[4];
                NMPC_UNROLL for (int jt = 0; jt < 4; jt++) {
                    M[jt] = mfma44(Y, X[jt], T(0));
                    // for the forward sweep: the same tile where the lane that needs it TRANSPOSED will read it
                    if (st_ok) lmk[TLM_MT + jt * 16 + tc * 4 + ta] = M[jt];
                }
                if (st_ok) lmk[TLM_Z + r] = mfma44(Y, (ta == tc) ? T(1) : T(0), T(0));   // Y' = L^-1 as a tile
                if (k > 0) {
                    // rows of Abar'(Pbar Abar): p rows copy, v rows add dt * p rows, q / omega rows are products
                    T qcol[4], qrow[4];
                    NMPC_UNROLL for (int t = 0; t < 4; t++) {      // unconditional reads + select: no exec-mask regions
                        const T qa = sh[natR[t] >= 0 ? natR[t] : 0], qb = sh[natC[t] >= 0 ? natC[t] : 0];
                        qcol[t] = (tc == 3 && natR[t] >= 0) ? qa : T(0);
                        qrow[t] = (ta == 3 && natC[t] >= 0) ? qb : T(0);
                    }
                    // Pbar_k = Qbar + Abar'(Pbar Abar) - Mbar'Mbar, assembled inside the matrix pipe: Qbar (stage
                    // Hessian diagonal, q_k in row / column 15) seeds the accumulators, -Mbar'Mbar is the last
                    // accumulation, so the result tiles never pass through the vector ALU.  (Element [15][15], the
                    // constant of the cost-to-go, just accumulates: nothing reads it.)
                    T Pn[4][4];
                    // Only the ten tiles on and above the diagonal are assembled and updated: Pbar is kept EXACTLY symmetric (the
                    // diagonal tiles averaged with their transposes, the tiles below as transposes - X' I transposes a tile).
                    // Computed independently, tile (i,j) and tile (j,i) differ by rounding, and that antisymmetric part is not
                    // contracted by the recursion: it grows by rho(A)^2 per stage (see team_as).
                    NMPC_UNROLL for (int jt = 0; jt < 4; jt++) {
                        T a2 = (jt == 2 ? Qdg[2] : T(0)) + (jt == 3 ? qcol[2] : T(0));
                        T a3 = (jt == 3 ? Qdg[3] + qcol[3] + qrow[3] : T(0));
                        if (jt >= 2) {
                            NMPC_UNROLL for (int kt = 0; kt < 4; kt++) {
                                if (kt < 3) a2 = mfma44(Aq0[kt], PA[kt][jt], a2);
                                if (jt == 3) a3 = mfma44(Aq1[kt], PA[kt][jt], a3);
                            }
                        }
                        Pn[0][jt] = PA[0][jt] + (jt == 0 ? Qdg[0] : T(0)) + (jt == 3 ? qcol[0] : T(0));
                        Pn[1][jt] = jt >= 1 ? dt_v * PA[0][jt] + PA[1][jt] + (jt == 1 ? Qdg[1] : T(0)) + (jt == 3 ? qcol[1] : T(0)) : T(0);
                        Pn[2][jt] = a2;
                        Pn[3][jt] = a3;
                    }
                    T Mn[4];
                    NMPC_UNROLL for (int t = 0; t < 4; t++) Mn[t] = -M[t];
                    NMPC_UNROLL for (int it = 0; it < 4; it++) {
                        NMPC_UNROLL for (int jt = it; jt < 4; jt++) Pt[it][jt] = mfma44(Mn[it], M[jt], Pn[it][jt]);
                    }
                    NMPC_UNROLL for (int it = 0; it < 4; it++) Pt[it][it] = T(0.5) * (Pt[it][it] + mfma44(Pt[it][it], Idt, T(0)));
                    NMPC_UNROLL for (int it = 1; it < 4; it++) {
                        NMPC_UNROLL for (int jt = 0; jt < it; jt++) Pt[it][jt] = mfma44(Pt[jt][it], Idt, T(0));
                    }
                    if (tP && k <= wnd && pol && st_ok) {
                        T *cp = tP + (size_t)k * TP_ROWS + r;
                        NMPC_UNROLL for (int it = 0; it < 4; it++) {
                            NMPC_UNROLL for (int jt = 0; jt < 4; jt++) cp[(it * 4 + jt) * 16] = Pt[it][jt];
                        }
                    }
                }
                NMPC_WSYNC();
            }
        }
        NMPC_STAMP(0)
        __syncthreads();   // L, m of every stage (written by lane 0) visible to the team
        bool pol_fail = false;
        if (act && !ok) {
            if (pol) pol_fail = true;                   // active-set pass (also one that produced a NaN, see team_as): give up this attempt
            else if (nanp) { status = 1; mode = M_DONE; }
            else { status = 4; mode = M_DONE; }
        }
        const bool act2 = mode != M_DONE, st_ok2 = act2 && valid;
        const bool ipm2 = act2 && ipm, pol2 = act2 && pol;

        // ================= sweep B: forward affine solve
        T xh = 0, rmax = T(1), s2 = 0;   // rmax: largest inverse step length, floor 1 => alpha_aff <= 1
        int p = 0;
        T nM[NU], nLf[10], nm[NU], n_uu = 0, n_l2 = 0, n_l3 = 0;
        auto prefetch_fwd = [&](int k) {
            const T *lmn = tLM + k * TLM_ROWS, *ivn = tIV + k * IV_ROWS;
            NMPC_UNROLL for (int i = 0; i < NU; i++) { nM[i] = lmn[rr * 4 + i]; nm[i] = lmn[62 + i]; }
            NMPC_UNROLL for (int i = 0; i < 10; i++) nLf[i] = lmn[52 + i];
            n_ul = NMPC_UL0( k * NU + j); n_pc = ivn[16 + j];
            if (any_ipm) { n_uu = ivn[j]; n_l2 = ivn[4 + j]; n_l3 = ivn[8 + j]; }
        };
        // active-set pass: with nothing pinned the KKT conditions reduce to "every input inside its box",
        // which this sweep sees by itself; only a pass with pins (or a violation to correct) needs sweep C
        // dirty: +1 per free input outside its box, +HEAVY per pinned input or NaN (a pass with only the
        // former is corrected without costates: sweep C-light)
        const T HEAVY = T(1048576);
        T dirty = 0;
        if constexpr (!MF) {
        prefetch_fwd(0);
        for (int k = 0; k < N; k++) {
            if (!SHARED) load_stage(k);
            T *ivk = tIV + k * IV_ROWS;
            T Lf[10], uh[NU];
            NMPC_UNROLL for (int i = 0; i < NU; i++) { sY[p * 64 + r * 4 + i] = nM[i] * xh; uh[i] = nm[i]; }
            NMPC_UNROLL for (int i = 0; i < 10; i++) Lf[i] = nLf[i];
            const T ul = n_ul, u = n_uu, ll = n_l2, lu = n_l3, pc = n_pc;
            sXh[p * 16 + r] = xh;
            if (k + 1 < N) prefetch_fwd(k + 1);
            NMPC_WSYNC();
            NMPC_UNROLL for (int i = 0; i < NU; i++) {
                T a = uh[i];
                NMPC_UNROLL for (int cc = 0; cc < NX; cc++) a += sY[p * 64 + cc * 4 + i];
                uh[i] = -a;
            }
            lt_solve(Lf, uh);
            {
                const T uj = sel4(uh, j);
                if (cmpl && st_ok2) ivk[12 + j] = uj;
                {
                    const T lo = lbj - ul, hi = ubj - ul;
                    const T tol = kkt_v * (T(1) + fabs(lo) + fabs(hi));
                    const bool clean = pc == T(0) && uj >= lo - tol && uj <= hi + tol;   // false for NaN
                    dirty += clean ? T(0) : ((pc == T(0) && uj == uj) ? T(1) : HEAVY);
                }
                if (any_ipm) {
                    const Pair<T> pr(u, ll, lu, lbj - ul, ubj - ul);
                    const T d = uj - u;
                    const T dla = -ll - pr.kl * d, dua = -lu + pr.ku * d;
                    // inverse step lengths: -d/tl, d/tu, -dla/ll = 1 + d/tl, -dua/lu = 1 - d/tu
                    const T a1 = d * pr.itl, a2 = d * pr.itu;
                    rmax = fmax(rmax, fmax(fmax(-a1, a2), fmax(T(1) + a1, T(1) - a2)));
                    s2 += dla * d - dua * d;
                }
            }
            if (pol2 && rowl && valid) tLM[k * TLM_ROWS + 66 + rr] = xh;   // xhat_k for the costate sweep
            {   // also through the last stage: the active-set check needs xhat_N
                T a = b_r;
                a += (rr < 3) ? xh + dt_v * sXh[p * 16 + rr + 3] : (rr < 6 ? xh : T(0));
                NMPC_UNROLL for (int cc = 0; cc < NZ; cc++) a += Adrow[cc] * sXh[p * 16 + 6 + cc];
                NMPC_UNROLL for (int i = 0; i < NU; i++) a += Brow[i] * uh[i];
                xh = a;
            }
            p ^= 1;
        }
        if (cmpl) { sRed[4 + j] = rmax; sRed[8 + j] = s2; sRed[j] = dirty; }
        } else {
            // ---- tile form (FP64): xbar = [xhat;1] lives as four column tiles (element 4t+a in lane (a,0));
            // v = Mbar xbar, u = -L^-T v and xbar+ = Abar xbar + B u are 17 v_mfma_f64_4x4x4 per stage
            // on transposed tiles (the A operand is read transposed), with no LDS exchange at all.
            const int ta = r >> 2, tc = r & 3;
            int natR[4], natC[4];
            NMPC_UNROLL for (int t = 0; t < 4; t++) { natR[t] = nat_of(t, ta); natC[t] = nat_of(t, tc); }
            const T lb_a = sel4(c.lbu, ta), ub_a = sel4(c.ubu, ta);
            T AT2[4], AT3[4], BT[4];
            auto load_tiles_T = [&]() {
                NMPC_UNROLL for (int it = 0; it < 4; it++) {
                    const int l = natC[it] >= 0 ? natC[it] : 0;
                    const bool real = natC[it] >= 0;
                    const T a2 = sAd[l * 8 + ta], a3 = sAd[l * 8 + 4 + (ta < 3 ? ta : 0)], bb = sB[l * 4 + ta], bv_ = sbv[l];
                    AT2[it] = real ? a2 : T(0);                                   // Abar[4it+c][8+a]
                    AT3[it] = real ? (ta < 3 ? a3 : bv_) : ((it == 3 && tc == 3 && ta == 3) ? T(1) : T(0));
                    BT[it] = real ? bb : T(0);                                    // B[4it+c][a]
                }
            };
            if (SHARED) load_tiles_T(); else fetch_stage(0);
            T xt[4];
            NMPC_UNROLL for (int t = 0; t < 4; t++) xt[t] = (t == 3 && ta == 3 && tc == 0) ? T(1) : T(0);
            // the tile path has no costate sweep: xhat is only read back when the caller wants the state trajectory
            const bool want_xhat = out.x_out != nullptr;
            const bool want_u = out.u_out != nullptr || out.x_out != nullptr || any_ipm;
            int kchgB = -1;           // highest stage whose pin set this pass changes
            const bool wave_pins = __ballot(pol2 && maybe_pins) != 0;
            int xslot[4];
            NMPC_UNROLL for (int t = 0; t < 4; t++) xslot[t] = natR[t] >= 0 ? natR[t] : 13;   // 66 + 13 = the pad slot
            // Per-stage operands (Mbar^T tiles, L^-1 tile, scalars of input a) come CHT stages at a time: a
            // stage is a short chain of dependent MFMAs, shorter than an L2 round trip.
            constexpr int CHT = 5;
            for (int k0 = 0; k0 < N; k0 += CHT) {
                T cMT[CHT][4], cZ[CHT], c_ul[CHT], c_pc[CHT], c_u[CHT], c_ll[CHT], c_lu[CHT], cG[CHT][5], c_rk[CHT];
                NMPC_UNROLL for (int i = 0; i < CHT; i++) {
                    const int k = (k0 + i < N) ? k0 + i : N - 1;
                    const T *lmn = tLM + k * TLM_ROWS, *ivn = tIV + k * IV_ROWS;
                    NMPC_UNROLL for (int g5 = 0; g5 < 5; g5++) cG[i][g5] = 0;
                    c_rk[i] = 0;
                    if (wave_pins) {               // gradient rows of the pinned inputs (stages without pins hold stale data)
                        NMPC_UNROLL for (int g5 = 0; g5 < 5; g5++) cG[i][g5] = lmn[TLM_G + g5 * 16 + r];
                        c_rk[i] = NMPC_TLD(w.qr, QRR, k * QR_ROWS + NX + ta);
                    }
                    NMPC_UNROLL for (int jt = 0; jt < 4; jt++) cMT[i][jt] = lmn[TLM_MT + jt * 16 + r];   // Mbar[c][4jt+a]
                    cZ[i] = lmn[TLM_Z + r];                                                            // (L^-1)[a][c]
                    c_ul[i] = NMPC_UL0( k * NU + ta); c_pc[i] = ivn[16 + ta];
                    c_u[i] = 0; c_ll[i] = 0; c_lu[i] = 0;
                    if (any_ipm) { c_u[i] = ivn[ta]; c_ll[i] = ivn[4 + ta]; c_lu[i] = ivn[8 + ta]; }
                }
                NMPC_UNROLL for (int i = 0; i < CHT; i++) {
                    const int k = k0 + i;
                    if (k < N) {
                        if (!SHARED) { put_stage(); if (k + 1 < N) fetch_stage(k + 1); load_tiles_T(); }
                        T *ivk = tIV + k * IV_ROWS;
                        const T ul = c_ul[i], pc = c_pc[i], u = c_u[i], ll = c_ll[i], lu = c_lu[i];
                        if (want_xhat && pol2 && valid && tc == 0) {   // xhat_k for the final sweep (pads land in the spare slot)
                            T *xs = tLM + k * TLM_ROWS + 66;
                            NMPC_UNROLL for (int t = 0; t < 4; t++) xs[xslot[t]] = xt[t];
                        }
                        // the part of Abar xbar that does not wait for u
                        T xn[4];
                        xn[0] = xt[0] + dt_v * xt[1]; xn[1] = xt[1]; xn[2] = 0; xn[3] = 0;
                        NMPC_UNROLL for (int it = 0; it < 4; it++) xn[it] = mfma44(AT3[it], xt[3], it < 3 ? mfma44(AT2[it], xt[2], xn[it]) : xn[it]);
                        // v = Mbar xbar (two chains), u = -L^-T v
                        const T v = mfma44(cMT[i][2], xt[2], mfma44(cMT[i][0], xt[0], T(0)))
                                  + mfma44(cMT[i][3], xt[3], mfma44(cMT[i][1], xt[1], T(0)));
                        const T ut = -mfma44(cZ[i], v, T(0));                         // lane (a,0): u_a
                        NMPC_UNROLL for (int it = 0; it < 4; it++) xn[it] = mfma44(BT[it], ut, xn[it]);
                        {
                            // KKT check of the pass, input a in lane (a,0): a free input must sit inside its box; a
                            // pinned one must have a multiplier of the right sign.  The costate is P x + p, so the
                            // gradient of a pinned input comes from the rows the factor sweep left for this stage
                            // and no adjoint sweep is needed; the corrected pin codes are written on the spot.
                            const T uj = ut;
                            const T lo = lb_a - ul, hi = ub_a - ul;
                            const T vpin = pc < T(0) ? lo : hi;
                            const bool pin_here = pol2 && pc != T(0);          // (interior-point teams keep other data there)
                            const T ue = pin_here ? vpin : uj;               // pinned inputs sit exactly on the bound
                            T npc;
                            bool nanq = !(uj == uj);
                            if (__ballot(tc == 0 && pin_here) != 0) {
                                const T uf = (tc == 0 && !pin_here) ? ue : T(0);                    // mask u
                                T g = mfma44(cG[i][4], uf, T(0));
                                T g2 = mfma44(cG[i][1], xt[1], mfma44(cG[i][0], xt[0], T(0)));
                                g = mfma44(cG[i][3], xt[3], mfma44(cG[i][2], xt[2], g));
                                g += g2 + sel4(c.Rd, ta) * ue + c_rk[i];
                                const T tolg = kkt_v * (T(1) + fabs(g));
                                const bool wrong = (pc < T(0) && g < -tolg) || (pc > T(0) && g > tolg);
                                const T tolb = kkt_v * (T(1) + fabs(lo) + fabs(hi));
                                npc = pin_here ? (wrong ? T(0) : pc) : (uj < lo - tolb ? T(-1) : (uj > hi + tolb ? T(1) : T(0)));
                                nanq |= !(g == g);
                            } else {
                                const T tolb = kkt_v * (T(1) + fabs(lo) + fabs(hi));
                                npc = uj < lo - tolb ? T(-1) : (uj > hi + tolb ? T(1) : T(0));
                            }
                            if (tc == 0) {
                                // candidate inputs: all stages for the interior-point sweeps and a requested input
                                // trajectory, stage 0 (the command u0) otherwise
                                if (st_ok2) { if (k == 0 || want_u) ivk[12 + ta] = ue; if (pol2 && npc != pc) ivk[16 + ta] = npc; }
                                dirty += nanq ? HEAVY : ((npc != pc) ? T(1) : T(0));
                                kchgB = (npc != pc) ? k : kchgB;                  // ascending k: the last one is the highest
                            }
                            if (any_ipm) {
                                const Pair<T> pr(u, ll, lu, lo, hi);
                                const T d = uj - u;
                                const T dla = -ll - pr.kl * d, dua = -lu + pr.ku * d;
                                const T a1 = d * pr.itl, a2 = d * pr.itu;
                                if (tc == 0) {
                                    rmax = fmax(rmax, fmax(fmax(-a1, a2), fmax(T(1) + a1, T(1) - a2)));
                                    s2 += dla * d - dua * d;
                                }
                            }
                        }
                        NMPC_UNROLL for (int t = 0; t < 4; t++) xt[t] = xn[t];
                    }
                }
            }
            // back to one natural row per lane for the sweeps that follow
            if (tc == 0) {
                NMPC_UNROLL for (int t = 0; t < 4; t++)
                    if (natR[t] >= 0) sXh[natR[t]] = xt[t];
            }
            NMPC_WSYNC();
            xh = sXh[rr];
            NMPC_WSYNC();
            if (tc == 0) { sRed[4 + ta] = rmax; sRed[8 + ta] = s2; sRed[ta] = dirty; sRed[28 + ta] = (T)kchgB; }
        }
        NMPC_STAMP(1)
        if (pol2 && rowl && valid) tLM[66 + rr] = xh;     // xhat_N parks in the unused xhat_0 slot (final sweep)
        __syncthreads();
        rmax = fmax(fmax(sRed[4], sRed[5]), fmax(sRed[6], sRed[7]));
        s2 = sRed[8] + sRed[9] + sRed[10] + sRed[11];
        dirty = sRed[0] + sRed[1] + sRed[2] + sRed[3] + ((xh == xh) ? T(0) : HEAVY);
        sXh[r] = dirty;                                   // a NaN in any xhat_N row marks the whole team
        __syncthreads();
        NMPC_UNROLL for (int l = 0; l < NX; l++) dirty += sXh[l];
        const bool unclean = pol2 && (pol_fail || !(dirty == T(0)));
        if (pol2 && !unclean) {                           // the pass satisfies the KKT conditions of the QP: done
            npol++;
            pass_in_attempt++;
            mode = M_DONE; from_ua = true; mu = 0; rho = 0;
        }
        if (MF && unclean) {                              // tile form: the forward sweep has already corrected the pins
            const int kc = (int)fmax(fmax(sRed[28], sRed[29]), fmax(sRed[30], sRed[31]));
            npol++;
            pass_in_attempt++;
            if (pol_fail || !(dirty < HEAVY) || pass_in_attempt >= c.polish_passes) { mode = M_IPM; pol_mu *= T(1e-2); }
            else { k_top = kc < ck_valid ? kc : N - 1; maybe_pins = true; }
        }
        const bool need_c = !MF && unclean;
        const bool heavy_c = need_c && (pol_fail || !(dirty < HEAVY));      // pins, NaN or a failed factorisation
        T sigmu;
        {
            const T aaff = T(1) / rmax;
            const T muaff = (T(1) - aaff) * mu + aaff * aaff * s2 / nc;
            T sg3 = muaff / mu;
            sg3 = sg3 * sg3 * sg3;
            sigmu = sg3 * mu;
        }

        // ================= sweep C (teams in an active-set pass): costates by the adjoint recursion,
        // KKT check of the pinned solve, corrected active set (primal-dual active-set step)
        if (__ballot(heavy_c) == 0 && __ballot(need_c) != 0) {
            // ---- sweep C-light: nothing was pinned, so there are no multipliers to check; the inputs that
            // left their box are pinned there and the pass repeats (same outcome as the full sweep C)
            T chg = 0;
            int kchg = -1;
            constexpr int CH = 10;
            for (int k0 = N - 1; k0 >= 0; k0 -= CH) {
                T c_ul[CH], c_uj[CH];
                NMPC_UNROLL for (int i = 0; i < CH; i++) {
                    const int k = (k0 - i > 0) ? k0 - i : 0;
                    c_ul[i] = NMPC_UL0( k * NU + j); c_uj[i] = tIV[k * IV_ROWS + 12 + j];
                }
                NMPC_UNROLL for (int i = 0; i < CH; i++) {
                    const int k = k0 - i;
                    if (k >= 0) {
                        const T lo = lbj - c_ul[i], hi = ubj - c_ul[i], uj = c_uj[i];
                        const T tol = kkt_v * (T(1) + fabs(lo) + fabs(hi));
                        const T npc = uj < lo - tol ? T(-1) : (uj > hi + tol ? T(1) : T(0));
                        chg += (npc != T(0)) ? T(1) : T(0);
                        kchg = (npc != T(0) && kchg < 0) ? k : kchg;
                        if (cmpl && need_c && valid) tIV[k * IV_ROWS + 16 + j] = npc;
                    }
                }
            }
            if (cmpl) { sRed[20 + j] = chg; sRed[28 + j] = (T)kchg; }
            __syncthreads();
            kchg = (int)fmax(fmax(sRed[28], sRed[29]), fmax(sRed[30], sRed[31]));
            if (need_c) {
                npol++;
                pass_in_attempt++;
                if (pass_in_attempt >= c.polish_passes) { mode = M_IPM; pol_mu *= T(1e-2); }
                else k_top = kchg < ck_valid ? kchg : N - 1;
            }
            __syncthreads();
        } else if (__ballot(need_c) != 0) {
            if (SHARED && MF) rows_from_lds();
            T pi_r = QdNr * xh + NMPC_TLD(w.qr, QRR, N * QR_ROWS + rr);   // xh = xhat_N after sweep B
            T chg = 0, nanf = (xh == xh) ? T(0) : T(1);
            int kchg = -1;           // highest stage whose pin set this check changes
            p = 0;
            constexpr int CH = 8;     // stage scalars are fetched a chunk at a time (see the final sweep)
            for (int k0 = N - 1; k0 >= 0; k0 -= CH) {
                T c_ul[CH], c_uj[CH], c_pc[CH], c_rk[CH], c_qr[CH], c_xk[CH];
                NMPC_UNROLL for (int i = 0; i < CH; i++) {
                    const int k = (k0 - i > 0) ? k0 - i : 0;
                    const T *ivk = tIV + k * IV_ROWS;
                    c_ul[i] = NMPC_UL0( k * NU + j); c_uj[i] = ivk[12 + j]; c_pc[i] = ivk[16 + j];
                    c_rk[i] = NMPC_TLD(w.qr, QRR, k * QR_ROWS + NX + j); c_qr[i] = NMPC_TLD(w.qr, QRR, k * QR_ROWS + rr);
                    c_xk[i] = tLM[k * TLM_ROWS + 66 + rr];
                }
                NMPC_UNROLL for (int i = 0; i < CH; i++) {
                    const int k = k0 - i;
                    if (k >= 0) {
                        if (!SHARED) load_stage(k);
                        T *ivk = tIV + k * IV_ROWS;
                        const T ul = c_ul[i], uj = c_uj[i], pc = c_pc[i], rk = c_rk[i], q_r = c_qr[i];
                        const T xk = k > 0 ? c_xk[i] : T(0);
                        sXh[p * 16 + r] = pi_r;
                        NMPC_WSYNC();
                        const T lo = lbj - ul, hi = ubj - ul;
                        const T vpin = pc < T(0) ? lo : hi;
                        const T ue = pc != T(0) ? vpin : uj;            // pinned inputs sit exactly on the bound
                        T g = Rdj * ue + rk, an = 0;
                        NMPC_UNROLL for (int l = 0; l < NX; l++) {
                            const T pl = sXh[p * 16 + l];
                            g += sB[l * 4 + j] * pl;
                            an += Acol[l] * pl;
                        }
                        T npc;
                        if (pc != T(0)) {                                // multiplier sign of a pinned input
                            const T tol = kkt_v *(T(1) + fabs(g));
                            const bool wrong = (pc < T(0) && g < -tol) || (pc > T(0) && g > tol);
                            npc = wrong ? T(0) : pc;
                        } else {                                         // free input inside its box?
                            const T tol = kkt_v *(T(1) + fabs(lo) + fabs(hi));
                            npc = uj < lo - tol ? T(-1) : (uj > hi + tol ? T(1) : T(0));
                        }
                        chg += (npc != pc) ? T(1) : T(0);
                        kchg = (npc != pc && kchg < 0) ? k : kchg;
                        nanf += (ue == ue && g == g && an == an) ? T(0) : T(1);
                        if (cmpl && need_c && valid) { ivk[16 + j] = npc; ivk[12 + j] = ue; }
                        pi_r = Qdr * xk + q_r + an;
                        p ^= 1;
                    }
                }
            }
            if (cmpl) { sRed[20 + j] = chg; sRed[24 + j] = nanf; sRed[28 + j] = (T)kchg; }
            __syncthreads();
            chg = sRed[20] + sRed[21] + sRed[22] + sRed[23];
            kchg = (int)fmax(fmax(sRed[28], sRed[29]), fmax(sRed[30], sRed[31]));
            nanf = sRed[24] + sRed[25] + sRed[26] + sRed[27];
            T xn = 0;                                            // NaN in any xhat row poisons the pass
            sXh[r] = nanf;
            __syncthreads();
            NMPC_UNROLL for (int l = 0; l < NX; l++) xn += sXh[l];
            if (need_c) {
                npol++;
                pass_in_attempt++;
                if (!pol_fail && xn == T(0) && chg == T(0)) { mode = M_DONE; from_ua = true; mu = 0; rho = 0; }
                else if (pol_fail || !(xn == T(0)) || pass_in_attempt >= c.polish_passes) { mode = M_IPM; pol_mu *= T(1e-2); }
                else k_top = kchg < ck_valid ? kchg : N - 1;
            }
            __syncthreads();
        }
        if (__ballot(ipm2) == 0) continue;   // nobody in this wave is iterating the interior point method

        // ================= sweep D: backward homogeneous solve
        T n_ua = 0;
        if constexpr (!MF) {
        pv = 0;
        p = 0;
        auto prefetch_bwd = [&](int k) {
            const T *lmn = tLM + k * TLM_ROWS, *ivn = tIV + k * IV_ROWS;
            NMPC_UNROLL for (int i = 0; i < NU; i++) nM[i] = lmn[rr * 4 + i];
            NMPC_UNROLL for (int i = 0; i < 10; i++) nLf[i] = lmn[52 + i];
            n_ul = NMPC_UL0( k * NU + j); n_uu = ivn[j]; n_l2 = ivn[4 + j]; n_l3 = ivn[8 + j]; n_ua = ivn[12 + j];
        };
        prefetch_bwd(N - 1);
        for (int k = N - 1; k >= 0; k--) {
            if (!SHARED) load_stage(k);
            T *lmk = tLM + k * TLM_ROWS;
            T Lf[10], Mc[NU];
            NMPC_UNROLL for (int i = 0; i < 10; i++) Lf[i] = nLf[i];
            NMPC_UNROLL for (int i = 0; i < NU; i++) Mc[i] = nM[i];
            {
                const T u = n_uu, ll = n_l2, lu = n_l3;
                const Pair<T> pr(u, ll, lu, lbj - n_ul, ubj - n_ul);
                const T da = n_ua - u;
                const T dla = -ll - pr.kl * da, dua = -lu + pr.ku * da;
                const T cl = dla * da, cu = -dua * da;
                if (cmpl) sDr[p * 4 + j] = -(sigmu - cl) * pr.itl + (sigmu - cu) * pr.itu;
            }
            sXh[p * 16 + r] = pv;
            if (k > 0) prefetch_bwd(k - 1);
            NMPC_WSYNC();
            T mv[NU];
            NMPC_UNROLL for (int i = 0; i < NU; i++) {
                T a = sDr[p * 4 + i];
                NMPC_UNROLL for (int l = 0; l < NX; l++) a += sB[l * 4 + i] * sXh[p * 16 + l];
                mv[i] = a;
            }
            l_solve(Lf, mv);
            if (r == 0 && ipm2 && valid) {
                NMPC_UNROLL for (int i = 0; i < NU; i++) lmk[62 + i] = mv[i];
            }
            if (k > 0) {
                T a = 0;
                NMPC_UNROLL for (int l = 0; l < NX; l++) a += Acol[l] * sXh[p * 16 + l];
                NMPC_UNROLL for (int i = 0; i < NU; i++) a -= Mc[i] * mv[i];
                pv = a;
            }
            p ^= 1;
        }
        } else {
            // ---- tile form (FP64): the costate of the correction is four column tiles; g = dr + B'pi, m = L^-1 g,
            // pi_k = Abar'pi - Mbar'm are 17 v_mfma_f64_4x4x4 per stage on the tiles the factor sweep left
            const int ta = r >> 2, tc = r & 3;
            int natR[4];
            NMPC_UNROLL for (int t = 0; t < 4; t++) natR[t] = nat_of(t, ta);
            const T lb_a = sel4(c.lbu, ta), ub_a = sel4(c.ubu, ta);
            T Aq0[4], Aq1z[4], Bt[4];
            auto load_tiles = [&]() {          // as in the factor sweep, without b and the homogeneous 1
                NMPC_UNROLL for (int kt = 0; kt < 4; kt++) {
                    const int l = natR[kt] >= 0 ? natR[kt] : 0;
                    const bool real = natR[kt] >= 0;
                    const T a0 = sAd[l * 8 + tc], a1 = sAd[l * 8 + 4 + (tc < 3 ? tc : 0)], bb = sB[l * 4 + tc];
                    Aq0[kt] = real ? a0 : T(0);
                    Aq1z[kt] = (real && tc < 3) ? a1 : T(0);
                    Bt[kt] = real ? bb : T(0);
                }
            };
            if (SHARED) load_tiles(); else fetch_stage(N - 1);
            T pit[4];
            NMPC_UNROLL for (int t = 0; t < 4; t++) pit[t] = 0;
            constexpr int CHD = 5;
            for (int k0 = N - 1; k0 >= 0; k0 -= CHD) {
                T cMn[CHD][4], cY[CHD], c_ul[CHD], c_u[CHD], c_ll[CHD], c_lu[CHD], c_ua[CHD];
                NMPC_UNROLL for (int i = 0; i < CHD; i++) {
                    const int k = (k0 - i > 0) ? k0 - i : 0;
                    const T *lmn = tLM + k * TLM_ROWS, *ivn = tIV + k * IV_ROWS;
                    NMPC_UNROLL for (int jt = 0; jt < 4; jt++) cMn[i][jt] = lmn[TLM_MT + jt * 16 + tc * 4 + ta];   // Mbar[a][4jt+c]
                    cY[i] = lmn[TLM_Z + tc * 4 + ta];                                                            // (L^-1)[c][a]
                    c_ul[i] = NMPC_UL0(k * NU + ta); c_u[i] = ivn[ta]; c_ll[i] = ivn[4 + ta]; c_lu[i] = ivn[8 + ta]; c_ua[i] = ivn[12 + ta];
                }
                NMPC_UNROLL for (int i = 0; i < CHD; i++) {
                    const int k = k0 - i;
                    if (k >= 0) {
                        if (!SHARED) { put_stage(); if (k > 0) fetch_stage(k - 1); load_tiles(); }
                        T *lmk = tLM + k * TLM_ROWS;
                        T drt;
                        {
                            const T u = c_u[i], ll = c_ll[i], lu = c_lu[i];
                            const Pair<T> pr(u, ll, lu, lb_a - c_ul[i], ub_a - c_ul[i]);
                            const T da = c_ua[i] - u;
                            const T dla = -ll - pr.kl * da, dua = -lu + pr.ku * da;
                            const T cl = dla * da, cu = -dua * da;
                            drt = tc == 0 ? -(sigmu - cl) * pr.itl + (sigmu - cu) * pr.itu : T(0);
                        }
                        const T g = mfma44(Bt[2], pit[2], mfma44(Bt[0], pit[0], drt)) + mfma44(Bt[3], pit[3], mfma44(Bt[1], pit[1], T(0)));
                        const T mvt = mfma44(cY[i], g, T(0));                     // lane (a,0): m_a
                        if (tc == 0 && ipm2 && valid) lmk[TLM_MT + 60 + ta] = mvt;  // column 15 of Mbar, where sweep E reads it
                        if (k > 0) {
                            T an[4];
                            an[0] = pit[0];
                            an[1] = dt_v * pit[0] + pit[1];
                            an[2] = mfma44(Aq0[2], pit[2], mfma44(Aq0[0], pit[0], T(0))) + mfma44(Aq0[1], pit[1], T(0));       // (Aq0[3] = 0)
                            an[3] = mfma44(Aq1z[2], pit[2], mfma44(Aq1z[0], pit[0], T(0))) + mfma44(Aq1z[3], pit[3], mfma44(Aq1z[1], pit[1], T(0)));
                            NMPC_UNROLL for (int t = 0; t < 4; t++) {
                                const T v = an[t] - mfma44(cMn[i][t], mvt, T(0));
                                pit[t] = (tc == 0 && natR[t] >= 0) ? v : T(0);
                            }
                        }
                    }
                }
            }
        }
        NMPC_STAMP(2)
        __syncthreads();   // m of every stage (written by lane 0) visible to the team

        // ================= sweep E: forward homogeneous solve, final direction, step length
        xh = 0;
        p = 0;
        rmax = c.tau;      // alpha = min(1, tau / max inverse step) = tau / max(rmax, tau)
        if constexpr (!MF) {
        auto prefetch_fwd2 = [&](int k) {
            prefetch_fwd(k);
            n_ua = tIV[k * IV_ROWS + 12 + j];
        };
        prefetch_fwd2(0);
        for (int k = 0; k < N; k++) {
            if (!SHARED) load_stage(k);
            T *ivk = tIV + k * IV_ROWS;
            T Lf[10], uh[NU];
            NMPC_UNROLL for (int i = 0; i < NU; i++) { sY[p * 64 + r * 4 + i] = nM[i] * xh; uh[i] = nm[i]; }
            NMPC_UNROLL for (int i = 0; i < 10; i++) Lf[i] = nLf[i];
            const T ul = n_ul, u = n_uu, ll = n_l2, lu = n_l3, ua = n_ua;
            sXh[p * 16 + r] = xh;
            if (k + 1 < N) prefetch_fwd2(k + 1);
            NMPC_WSYNC();
            NMPC_UNROLL for (int i = 0; i < NU; i++) {
                T a = uh[i];
                NMPC_UNROLL for (int cc = 0; cc < NX; cc++) a += sY[p * 64 + cc * 4 + i];
                uh[i] = -a;
            }
            lt_solve(Lf, uh);
            {
                const Pair<T> pr(u, ll, lu, lbj - ul, ubj - ul);
                const T da = ua - u;
                const T dla = -ll - pr.kl * da, dua = -lu + pr.ku * da;
                const T cl = dla * da, cu = -dua * da;
                const T d = da + sel4(uh, j);
                if (cmpl && ipm2 && valid) ivk[16 + j] = d;
                const T dl = -ll - (cl - sigmu) * pr.itl - pr.kl * d;
                const T du = -lu - (cu - sigmu) * pr.itu + pr.ku * d;
                rmax = fmax(rmax, fmax(-d * pr.itl, d * pr.itu));
                rmax = fmax(rmax, fmax(-dl * fast_rcp(ll), -du * fast_rcp(lu)));
            }
            if (k < N - 1) {
                T a = 0;
                a += (rr < 3) ? xh + dt_v * sXh[p * 16 + rr + 3] : (rr < 6 ? xh : T(0));
                NMPC_UNROLL for (int cc = 0; cc < NZ; cc++) a += Adrow[cc] * sXh[p * 16 + 6 + cc];
                NMPC_UNROLL for (int i = 0; i < NU; i++) a += Brow[i] * uh[i];
                xh = a;
            }
            p ^= 1;
        }
        if (cmpl) sRed[12 + j] = rmax;
        } else {
            // ---- tile form (FP64): as sweep B, homogeneous dynamics (no b), m from sweep D in column 15 of Mbar
            const int ta = r >> 2, tc = r & 3;
            int natC[4];
            NMPC_UNROLL for (int t = 0; t < 4; t++) natC[t] = nat_of(t, tc);
            const T lb_a = sel4(c.lbu, ta), ub_a = sel4(c.ubu, ta);
            T AT2[4], AT3z[4], BT[4];
            auto load_tiles_T = [&]() {
                NMPC_UNROLL for (int it = 0; it < 4; it++) {
                    const int l = natC[it] >= 0 ? natC[it] : 0;
                    const bool real = natC[it] >= 0;
                    const T a2 = sAd[l * 8 + ta], a3 = sAd[l * 8 + 4 + (ta < 3 ? ta : 0)], bb = sB[l * 4 + ta];
                    AT2[it] = real ? a2 : T(0);
                    AT3z[it] = (real && ta < 3) ? a3 : T(0);
                    BT[it] = real ? bb : T(0);
                }
            };
            if (SHARED) load_tiles_T(); else fetch_stage(0);
            T xt[4];
            NMPC_UNROLL for (int t = 0; t < 4; t++) xt[t] = 0;
            const T one15 = (ta == 3 && tc == 0) ? T(1) : T(0);        // homogeneous coordinate, for the m term only
            constexpr int CHE = 5;
            for (int k0 = 0; k0 < N; k0 += CHE) {
                T cMT[CHE][4], cZ[CHE], c_ul[CHE], c_u[CHE], c_ll[CHE], c_lu[CHE], c_ua[CHE];
                NMPC_UNROLL for (int i = 0; i < CHE; i++) {
                    const int k = (k0 + i < N) ? k0 + i : N - 1;
                    const T *lmn = tLM + k * TLM_ROWS, *ivn = tIV + k * IV_ROWS;
                    NMPC_UNROLL for (int jt = 0; jt < 4; jt++) cMT[i][jt] = lmn[TLM_MT + jt * 16 + r];
                    cZ[i] = lmn[TLM_Z + r];
                    c_ul[i] = NMPC_UL0(k * NU + ta); c_u[i] = ivn[ta]; c_ll[i] = ivn[4 + ta]; c_lu[i] = ivn[8 + ta]; c_ua[i] = ivn[12 + ta];
                }
                NMPC_UNROLL for (int i = 0; i < CHE; i++) {
                    const int k = k0 + i;
                    if (k < N) {
                        if (!SHARED) { put_stage(); if (k + 1 < N) fetch_stage(k + 1); load_tiles_T(); }
                        T *ivk = tIV + k * IV_ROWS;
                        T xn[4];
                        xn[0] = xt[0] + dt_v * xt[1]; xn[1] = xt[1]; xn[2] = 0; xn[3] = 0;
                        NMPC_UNROLL for (int it = 0; it < 4; it++) xn[it] = mfma44(AT3z[it], xt[3], it < 3 ? mfma44(AT2[it], xt[2], xn[it]) : xn[it]);
                        const T v = mfma44(cMT[i][2], xt[2], mfma44(cMT[i][0], xt[0], T(0)))
                                  + mfma44(cMT[i][3], xt[3] + one15, mfma44(cMT[i][1], xt[1], T(0)));
                        const T ut = -mfma44(cZ[i], v, T(0));
                        NMPC_UNROLL for (int it = 0; it < 4; it++) xn[it] = mfma44(BT[it], ut, xn[it]);
                        {
                            const T u = c_u[i], ll = c_ll[i], lu = c_lu[i], ua = c_ua[i], ul = c_ul[i];
                            const Pair<T> pr(u, ll, lu, lb_a - ul, ub_a - ul);
                            const T da = ua - u;
                            const T dla = -ll - pr.kl * da, dua = -lu + pr.ku * da;
                            const T cl = dla * da, cu = -dua * da;
                            const T d = da + ut;
                            if (tc == 0 && ipm2 && valid) ivk[16 + ta] = d;
                            const T dl = -ll - (cl - sigmu) * pr.itl - pr.kl * d;
                            const T du = -lu - (cu - sigmu) * pr.itu + pr.ku * d;
                            if (tc == 0) {
                                rmax = fmax(rmax, fmax(-d * pr.itl, d * pr.itu));
                                rmax = fmax(rmax, fmax(-dl * fast_rcp(ll), -du * fast_rcp(lu)));
                            }
                        }
                        NMPC_UNROLL for (int t = 0; t < 4; t++) xt[t] = xn[t];
                    }
                }
            }
            if (tc == 0) sRed[12 + ta] = rmax;
        }
        NMPC_STAMP(3)
        __syncthreads();
        rmax = fmax(fmax(sRed[12], sRed[13]), fmax(sRed[14], sRed[15]));
        const T alpha = c.tau / rmax;
        // ================= sweep F: primal-dual update, duality measure of the new iterate
        T ms = 0;
        constexpr int CHF = 10;           // iterate and directions are fetched a chunk of stages at a time (see sweep C)
        for (int k0 = 0; k0 < N; k0 += CHF) {
            T f_ul[CHF], f_u[CHF], f_ll[CHF], f_lu[CHF], f_ua[CHF], f_d[CHF];
            NMPC_UNROLL for (int i = 0; i < CHF; i++) {
                const int k = (k0 + i < N) ? k0 + i : N - 1;
                const T *ivn = tIV + k * IV_ROWS;
                f_ul[i] = NMPC_UL0(k * NU + j);
                f_u[i] = ivn[j]; f_ll[i] = ivn[4 + j]; f_lu[i] = ivn[8 + j]; f_ua[i] = ivn[12 + j]; f_d[i] = ivn[16 + j];
            }
            NMPC_UNROLL for (int i = 0; i < CHF; i++) {
                const int k = k0 + i;
                if (k < N) {
                    T *ivk = tIV + k * IV_ROWS;
                    T u = f_u[i], ll = f_ll[i], lu = f_lu[i];
                    const T lo = lbj - f_ul[i], hi = ubj - f_ul[i];
                    const Pair<T> pr(u, ll, lu, lo, hi);
                    const T da = f_ua[i] - u, d = f_d[i];
                    const T dla = -ll - pr.kl * da, dua = -lu + pr.ku * da;
                    const T cl = dla * da, cu = -dua * da;
                    const T dl = -ll - (cl - sigmu) * pr.itl - pr.kl * d;
                    const T du = -lu - (cu - sigmu) * pr.itu + pr.ku * d;
                    u += alpha * d; ll += alpha * dl; lu += alpha * du;
                    if (cmpl && ipm2 && valid) {
                        ivk[j] = u; ivk[4 + j] = ll; ivk[8 + j] = lu;
                        // active-set guess for a later polish attempt: a bound whose multiplier exceeds its slack
                        ivk[16 + j] = ll > u - lo ? T(-1) : (lu > hi - u ? T(1) : T(0));
                    }
                    ms += ll * (u - lo) + lu * (hi - u);
                }
            }
        }
        NMPC_STAMP(4)
        if (cmpl) sRed[16 + j] = ms;
        __syncthreads();
        ms = sRed[16] + sRed[17] + sRed[18] + sRed[19];
        if (ipm2) {
            if (!(alpha == alpha)) { status = 1; mode = M_DONE; }
            else if (alpha < T(1e-12)) { status = 3; mode = M_DONE; }
            else {
                rho *= (T(1) - alpha);
                mu = ms / nc;
            }
        }
        __syncthreads();   // sRed is reused by the next iteration
    }

    NMPC_STAMP(5)
    // ---- final sweep: state rollout from the inputs, full SQP step (U1)
    T u0_new = 0;
    bool outputs_done = false;
    {
        T dx = 0;
        bool bad = false;
        int p = 0;
        const bool upd = (status == 0 || status == 2);
        // loads of a chunk of stages are issued together: the read-modify-write of xl/ul may alias as
        // far as the compiler knows, and one exposed global round trip per stage dominated this sweep
        constexpr int CH = 8;
        const int uoff = (from_ua ? 12 : 0) + j;
        // an instance that ended on an accepted active-set pass: its forward sweep already left xhat_k (the same
        // recursion on the same inputs), so the step is applied stage-parallel.  The choice is per TEAM - taken per wave
        // (all four instances accepted, or the rollout below for everybody) it made the last bits of an accepted
        // instance's state trajectory depend on how its wave-mates ended (found by tools/dev/fuzz_perm.py)
        const bool fast = from_ua;
        const bool any_fast = __ballot(valid && from_ua) != 0, any_slow = __ballot(valid && !from_ua) != 0;
        if (any_fast) {
            // the workspace iterate (xl, ul) is not read again after this kernel: the new iterate goes
            // straight to the caller's arrays (13- and 4-element runs per team), or nowhere but u0
            if (fast) u0_new = NMPC_UL0( j) + tIV[uoff];
            if (out.x_out || out.u_out) {
                for (int k0 = 0; k0 <= N; k0 += CH) {
                    T uv[CH], ulv[CH], xlv[CH], xhv[CH];
                    NMPC_UNROLL for (int i = 0; i < CH; i++) {
                        const int k = (k0 + i <= N) ? k0 + i : N, ku = k < N ? k : N - 1;
                        uv[i] = tIV[ku * IV_ROWS + uoff];
                        ulv[i] = NMPC_UL0( ku * NU + j);
                        xlv[i] = NMPC_TLD(w.xl, XLR, (SHARED ? 0 : k) * NX + rr);        // shared cold start: x_k = x0, staged once
                        xhv[i] = tLM[(k < N ? k * TLM_ROWS : 0) + 66 + rr];      // xhat_N sits in the stage-0 slot
                    }
                    NMPC_UNROLL for (int i = 0; i < CH; i++) {
                        const int k = k0 + i;
                        if (k <= N && valid && fast) {
                            if (out.x_out && rowl)
                                out.x_out[((size_t)inst * (N + 1) + k) * NX + rr] = xlv[i] + (k > 0 ? xhv[i] : T(0));
                            if (out.u_out && cmpl && k < N) out.u_out[((size_t)inst * N + k) * NU + j] = ulv[i] + uv[i];
                        }
                    }
                }
            }
            outputs_done = fast;
        }
        if (any_slow) {
        if (SHARED && MF) rows_from_lds();
        for (int k0 = 0; k0 < N; k0 += CH) {
            T uv[CH], ulv[CH], xlv[CH];
            NMPC_UNROLL for (int i = 0; i < CH; i++) {
                const int k = (k0 + i < N) ? k0 + i : N - 1;
                uv[i] = tIV[k * IV_ROWS + uoff];
                ulv[i] = NMPC_UL0( k * NU + j);
                xlv[i] = NMPC_TLD(w.xl, XLR, (SHARED ? 0 : k + 1) * NX + rr);
            }
            NMPC_UNROLL for (int i = 0; i < CH; i++) {
                const int k = k0 + i;
                if (k < N) {
                    if (!SHARED) load_stage(k);
                    const T u = uv[i];
                    if (cmpl) sDr[p * 4 + j] = u;
                    sXh[p * 16 + r] = dx;
                    NMPC_WSYNC();
                    T du[NU];
                    NMPC_UNROLL for (int ii = 0; ii < NU; ii++) { du[ii] = sDr[p * 4 + ii]; bad |= !(du[ii] == du[ii]); }
                    T a = b_r;
                    a += (rr < 3) ? dx + dt_v * sXh[p * 16 + rr + 3] : (rr < 6 ? dx : T(0));
                    NMPC_UNROLL for (int cc = 0; cc < NZ; cc++) a += Adrow[cc] * sXh[p * 16 + 6 + cc];
                    NMPC_UNROLL for (int ii = 0; ii < NU; ii++) a += Brow[ii] * du[ii];
                    dx = a;
                    if (upd && valid && !fast) {
                        if (cmpl) NMPC_TST(w.ul, ULR, k * NU + j, ulv[i] + u);
                        if (rowl) NMPC_TST(w.xl, XLR, (k + 1) * NX + rr, xlv[i] + dx);
                    }
                    p ^= 1;
                }
            }
        }
        }
        if (any_slow) {
            // NaN anywhere in the step poisons the instance: reduce the flag over the team
            // (an accepted active-set pass has been checked already)
            sXh[r] = (dx == dx && !bad) ? T(0) : T(1);
            __syncthreads();
            T nb = 0;
            NMPC_UNROLL for (int l = 0; l < NX; l++) nb += sXh[l];
            if (!fast) {
                if (nb > T(0) && upd) status = 1;
                u0_new = NMPC_TLD(w.ul, ULR, j);
            }
        }
    }
    NMPC_STAMP(6)
    NMPC_PROF_END(w)
    const int nlp_status = (status == 2) ? 0 : (status == 3 ? 4 : status);
    if (valid) {
        if (r == 0 && out.status) out.status[inst] = nlp_status;
        if (r == 0) { w.iters[inst] = it; w.status[inst] = nlp_status; if (w.npol) w.npol[inst] = from_ua ? npol : -npol; }   // > 0: accepted active-set solution
        if (cmpl) out.u0[(size_t)inst * NU + j] = nlp_status == 0 ? u0_new : T(0);   // controller.py:448-452
        if (outputs_done) return;
        // A failed instance (status 1 / 4) hands back the cold-start point (x_k = x0, u_k = 0,
        // controller.py:425-431) instead of its - possibly poisoned, and under the shared cold start never
        // staged - linearisation point: a caller that feeds x_out / u_out back as the next warm start then
        // restarts cold, which is what the reference does after a failure (controller.py:448-450).
        const bool failed = nlp_status != 0;
        const T x0r = NMPC_TLD(w.xl, XLR, rr);                 // stage 0 is pinned to x0 in every variant
        if (out.x_out && rowl) {
            for (int k = 0; k <= N; k++)
                out.x_out[((size_t)inst * (N + 1) + k) * NX + rr] = failed ? x0r : NMPC_TLD(w.xl, XLR, k * NX + rr);
        }
        if (out.u_out && cmpl) {
            for (int k = 0; k < N; k++) out.u_out[((size_t)inst * N + k) * NU + j] = failed ? T(0) : NMPC_TLD(w.ul, ULR, k * NU + j);
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// Tile form of the model Jacobian (controller.py:267-355) for the forward sensitivities.  In the padded
// order p_|v_|q|w_ the Jacobian has five non-zero 4x4 tiles: (p,v) = I, (v,q) = Fvq, (q,q) = Fqq,
// (q,w) = Fqw, (w,w) = Fww.  Every entry is +-(0.5|1|2|k) times ONE state component, so lane (a,c) of a
// team forms its element of each TRANSPOSED tile (what the MFMA wants as its A operand) as a dot product
// of the state with a constant coefficient pattern; the patterns below are indexed by r = 4a + c and
// restate model_jac() entry by entry (checked against it by the GPU parity tests).
static __device__ const signed char NMPC_TVQ[16][4] = {   // Fvq[c][a] / t2 over (qw,qx,qy,qz)
    {0, 0, 1, 0}, {0, -1, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0},
    {0, 0, 0, 1}, {-1, 0, 0, 0}, {0, -2, 0, 0}, {0, 0, 0, 0},
    {1, 0, 0, 0}, {0, 0, 0, 1}, {0, 0, -2, 0}, {0, 0, 0, 0},
    {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
static __device__ const signed char NMPC_TQQ[16][3] = {   // Fqq[c][a] / 0.5 over (wx,wy,wz)
    {0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1},
    {-1, 0, 0}, {0, 0, 0}, {0, 0, -1}, {0, 1, 0},
    {0, -1, 0}, {0, 0, 1}, {0, 0, 0}, {-1, 0, 0},
    {0, 0, -1}, {0, -1, 0}, {1, 0, 0}, {0, 0, 0}};
static __device__ const signed char NMPC_TQW[16][4] = {   // Fqw[c][a] / 0.5 over (qw,qx,qy,qz), a < 3
    {0, -1, 0, 0}, {1, 0, 0, 0}, {0, 0, 0, 1}, {0, 0, -1, 0},
    {0, 0, -1, 0}, {0, 0, 0, -1}, {1, 0, 0, 0}, {0, 1, 0, 0},
    {0, 0, 0, -1}, {0, 0, 1, 0}, {0, -1, 0, 0}, {1, 0, 0, 0},
    {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
static __device__ const signed char NMPC_TWW[16][3] = {   // Fww[c][a] / k_c over (wx,wy,wz), a, c < 3
    {0, 0, 0}, {0, 0, 1}, {0, 1, 0}, {0, 0, 0},
    {0, 0, 1}, {0, 0, 0}, {1, 0, 0}, {0, 0, 0},
    {0, 1, 0}, {1, 0, 0}, {0, 0, 0}, {0, 0, 0},
    {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};

template <class T>
struct JacCoef {     // per-lane constants of the tile Jacobian
    T vq[4], qq[3], qw[4], ww[3];   // coefficient patterns (ww already times k_c)
    T fu;                           // f_u tile of the omega rows: fuw[a][c], a < 3
};

template <class T>
__device__ __forceinline__ void jac_coef(const Consts<T> &c, int r, JacCoef<T> &k)
{
    const int ta = r >> 2, tc = r & 3;
    NMPC_UNROLL for (int i = 0; i < 4; i++) { k.vq[i] = (T)NMPC_TVQ[r][i]; k.qw[i] = T(0.5) * (T)NMPC_TQW[r][i]; }
    T kx = -(c.J[2] - c.J[1]) * c.invJ[0], ky = -(c.J[0] - c.J[2]) * c.invJ[1], kz = -(c.J[1] - c.J[0]) * c.invJ[2];
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(kx), "+v"(ky), "+v"(kz));      // three values and two selects - left alone, the compiler builds an if / else per lane
#endif
    const T kc = tc == 0 ? kx : (tc == 1 ? ky : kz);
    NMPC_UNROLL for (int i = 0; i < 3; i++) { k.qq[i] = T(0.5) * (T)NMPC_TQQ[r][i]; k.ww[i] = kc * (T)NMPC_TWW[r][i]; }
    T fu = 0;
    NMPC_UNROLL for (int i = 0; i < 3; i++) {
        NMPC_UNROLL for (int jj = 0; jj < NU; jj++) fu = (i == ta && jj == tc) ? c.fuw[i][jj] : fu;
    }
    k.fu = fu;
}

// the four transposed Jacobian tiles and the thrust-direction entry r3m[a] at state x, input u
template <class T>
__device__ __forceinline__ void jac_tiles(const Consts<T> &c, const JacCoef<T> &k, const T *x, const T *u, int ta,
                                          T &FvqT, T &FqqT, T &FqwT, T &FwwT, T &r3a)
{
    const T qw = x[6], qx = x[7], qy = x[8], qz = x[9], wx = x[10], wy = x[11], wz = x[12];
    const T t2 = T(2) * (u[0] + u[1] + u[2] + u[3]) * c.inv_mass;
    FvqT = t2 * (k.vq[0] * qw + k.vq[1] * qx + k.vq[2] * qy + k.vq[3] * qz);
    FqqT = k.qq[0] * wx + k.qq[1] * wy + k.qq[2] * wz;
    FqwT = k.qw[0] * qw + k.qw[1] * qx + k.qw[2] * qy + k.qw[3] * qz;
    FwwT = k.ww[0] * wx + k.ww[1] * wy + k.ww[2] * wz;
    const T r0 = T(2) * (qx * qz + qw * qy) * c.inv_mass, r1 = T(2) * (qy * qz - qw * qx) * c.inv_mass,
            r2 = (T(1) - T(2) * (qx * qx + qy * qy)) * c.inv_mass;
    r3a = ta == 0 ? r0 : (ta == 1 ? r1 : (ta == 2 ? r2 : T(0)));
}

// K = J S + [0 | f_u]: S, K are 4 row tiles (p_, v_, q, w_) x 3 column tiles (q | w_ | u)
// ZWQ: the caller keeps tile (3,0) - d omega / d q, identically zero - out of its updates: its products are skipped
template <class T, bool ZWQ = false>
__device__ __forceinline__ void vde_tiles(T FvqT, T FqqT, T FqwT, T FwwT, T r3a, T fu, const T S[4][3], T K[4][3])
{
    NMPC_UNROLL for (int ct = 0; ct < 3; ct++) {
        K[0][ct] = S[1][ct];
        K[1][ct] = mfma44(FvqT, S[2][ct], ct == 2 ? r3a : T(0));
        if (ZWQ && ct == 0) {
            K[2][ct] = mfma44(FqqT, S[2][ct], T(0));
            K[3][ct] = T(0);
        } else {
            K[2][ct] = mfma44(FqwT, S[3][ct], mfma44(FqqT, S[2][ct], T(0)));
            K[3][ct] = mfma44(FwwT, S[3][ct], ct == 2 ? fu : T(0));
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Preparation phase in the team mapping (staging of controller.py:414-445 + linearisation, U2/U3):
// lane r stages row r of every stage (coalesced 13- and 17-element runs of x0 / x_init / yref), and
// lane c < 11 integrates column c of the forward sensitivities (4 quaternion, 3 body-rate, 4 input
// columns) with the explicit-midpoint steps; the state itself and the four Jacobians are replicated
// (cheap).  Writes the SoA rows the QP kernel reads (xl, ul, qr) and the per-instance stage block tAB.
template <class T>
__device__ __forceinline__ void vde_col_rt(const Consts<T> &c, const Jac<T> &J, const T *s, T *k, bool is_u, int ju)
{
    NMPC_UNROLL for (int i = 0; i < 3; i++) k[i] = s[3 + i];
    NMPC_UNROLL for (int i = 0; i < 3; i++) {
        T a = is_u ? J.r3m[i] : T(0);
        NMPC_UNROLL for (int l = 0; l < 4; l++) a += J.Fvq[i][l] * s[6 + l];
        k[3 + i] = a;
    }
    NMPC_UNROLL for (int i = 0; i < 4; i++) {
        T a = 0;
        NMPC_UNROLL for (int l = 0; l < 4; l++) a += J.Fqq[i][l] * s[6 + l];
        NMPC_UNROLL for (int l = 0; l < 3; l++) a += J.Fqw[i][l] * s[10 + l];
        k[6 + i] = a;
    }
    NMPC_UNROLL for (int i = 0; i < 3; i++) {
        T a = is_u ? sel4(c.fuw[i], ju) : T(0);
        NMPC_UNROLL for (int l = 0; l < 3; l++) a += J.Fww[i][l] * s[10 + l];
        k[10 + i] = a;
    }
}

// TI: element type of the caller's input arrays (float for NMPC_DTYPE_F32IO)
template <class T, class TI = T>
__device__ __forceinline__ void team_prepare(const Consts<T> &c, const Work<T> &w, const Inputs<TI> &in, int B, int tpw,
                                             T *smem = nullptr, int inst_ov = -2)
{
    // smem != null (fused launch) and a shared cold-start linearisation: the stage matrices go straight
    // into the team's LDS arrays (natural layout, as load_stage leaves them) and never touch HBM
    const int tid = threadIdx.x, team = (tid >> 2) & 3, r = ((tid >> 4) << 2) | (tid & 3);   // as team_ipm
    const int rr = r < NX ? r : NX - 1, j = r & 3;
    const bool rowl = r < NX, cmpl = r < NU;
    int inst = inst_ov != -2 ? inst_ov : blockIdx.x * tpw + team;
    const bool valid = inst_ov != -2 ? (inst >= 0 && inst < B) : (team < tpw && inst < B);
    if (!valid) inst = B - 1;
    const int N = c.N;
    const int XLR = (N + 1) * NX, ULR = N * NU, QRR = N * QR_ROWS + NX;
    const bool warm = in.x_init != nullptr && in.u_init != nullptr;
    const TI *x0 = in.x0 + (size_t)inst * NX;
    const TI *yr = in.yref_bcast ? in.yref : in.yref + (size_t)inst * N * NY;
    const TI *ye = in.yref_bcast ? in.yref_e : in.yref_e + (size_t)inst * NX;
    const TI *xi = warm ? in.x_init + (size_t)inst * (N + 1) * NX : nullptr;
    const TI *ui = warm ? in.u_init + (size_t)inst * N * NU : nullptr;
    const T Wqr = [&] { T v = 0; NMPC_UNROLL for (int i = 0; i < NX; i++) v = (i == rr) ? c.Wq[i] : v; return v; }();
    const T WqNr = [&] { T v = 0; NMPC_UNROLL for (int i = 0; i < NX; i++) v = (i == rr) ? c.WqN[i] : v; return v; }();
    const T Wrj = sel4(c.Wr, j);
    // staging and cost gradients (U4).  The inputs of the first HC stages are fetched BEFORE the
    // linearisation so that its arithmetic hides their latency; every later load is issued a chunk of
    // stages at a time before any store (the workspace stores may alias the inputs as far as the
    // compiler knows, and a load-store-load chain per stage costs one HBM round trip each).
    const T x0r = x0[rr];
    constexpr int HC = 20, CH = 8;
    T hx[HC], hu[HC], hyx[HC], hyu[HC];
    if (warm) {                      // one wave-uniform branch instead of a select per stage and array
        NMPC_UNROLL for (int i = 0; i < HC; i++) {
            const int k = i < N ? i : N - 1;
            hx[i] = k > 0 ? xi[(size_t)k * NX + rr] : x0r;              // stage 0 is pinned to x0
            hu[i] = ui[(size_t)k * NU + j];
            hyx[i] = yr[(size_t)k * NY + rr];
            hyu[i] = yr[(size_t)k * NY + NX + j];
        }
    } else {
        NMPC_UNROLL for (int i = 0; i < HC; i++) {
            const int k = i < N ? i : N - 1;
            hx[i] = x0r;
            hu[i] = T(0);
            hyx[i] = yr[(size_t)k * NY + rr];
            hyu[i] = yr[(size_t)k * NY + NX + j];
        }
    }
    const T xN = warm ? xi[(size_t)N * NX + rr] : x0r, yeN = ye[rr];
    // linearisation: one interval if the cold start lets all stages share it
    const int Ns = c.shared ? 1 : N;
    const int col = r < 11 ? r : 10;
    const bool is_u = col >= 7;
    if constexpr (sizeof(T) == 8) {
        // ---- tile form (FP64): the sensitivities are 4 x 3 register tiles, each VDE evaluation is 12
        // v_mfma_f64_4x4x4 plus the four Jacobian tiles; no per-lane column logic, no Jacobian struct
        const int ta = r >> 2, tc = r & 3;
        int natR[4];
        NMPC_UNROLL for (int t = 0; t < 4; t++) natR[t] = nat_of(t, ta);
        JacCoef<T> jk;
        jac_coef(c, r, jk);
        const T hh = T(0.5) * c.h;
        for (int k = 0; k < Ns; k++) {
            T xs[NX], us[NU], S[4][3];
            NMPC_UNROLL for (int i = 0; i < NX; i++) xs[i] = (warm && k > 0) ? xi[(size_t)k * NX + i] : x0[i];
            NMPC_UNROLL for (int i = 0; i < NU; i++) us[i] = warm ? ui[(size_t)k * NU + i] : T(0);
            NMPC_UNROLL for (int rt = 0; rt < 4; rt++) {
                NMPC_UNROLL for (int ct = 0; ct < 3; ct++) S[rt][ct] = 0;
            }
            S[2][0] = (ta == tc) ? T(1) : T(0);                  // d q / d q = I
            S[3][1] = (ta == tc && ta < 3) ? T(1) : T(0);        // d w / d w = I
            for (int st = 0; st < c.steps; st++) {
                T f1[NX], xm[NX], f2[NX], K[4][3], Sm[4][3];
                T a1, a2, a3, a4, a5;
                model_f(c, xs, us, f1);
                jac_tiles(c, jk, xs, us, ta, a1, a2, a3, a4, a5);
                vde_tiles(a1, a2, a3, a4, a5, jk.fu, S, K);
                NMPC_UNROLL for (int i = 0; i < NX; i++) xm[i] = xs[i] + hh * f1[i];
                NMPC_UNROLL for (int rt = 0; rt < 4; rt++) {
                    NMPC_UNROLL for (int ct = 0; ct < 3; ct++) Sm[rt][ct] = S[rt][ct] + hh * K[rt][ct];
                }
                model_f(c, xm, us, f2);
                jac_tiles(c, jk, xm, us, ta, a1, a2, a3, a4, a5);
                vde_tiles(a1, a2, a3, a4, a5, jk.fu, Sm, K);
                NMPC_UNROLL for (int i = 0; i < NX; i++) xs[i] += c.h * f2[i];
                NMPC_UNROLL for (int rt = 0; rt < 4; rt++) {
                    NMPC_UNROLL for (int ct = 0; ct < 3; ct++) S[rt][ct] += c.h * K[rt][ct];
                }
            }
            const bool to_lds = smem != nullptr && c.shared;
            if (valid || to_lds) {     // natural layout: Ad rows [13][8] | B rows [13][4] | b [13]
                T *a = to_lds ? smem + team * TEAM_LDS : w.tAB + ((size_t)inst * Ns + k) * TAB_ROWS;
                const int oA = to_lds ? L_AD : 0, oB = to_lds ? L_B : 104, ob = to_lds ? L_BV : 156;
                NMPC_UNROLL for (int rt = 0; rt < 4; rt++) {
                    if (natR[rt] >= 0) {
                        a[oA + natR[rt] * 8 + tc] = S[rt][0];
                        a[oA + natR[rt] * 8 + 4 + tc] = tc < 3 ? S[rt][1] : T(0);
                        a[oB + natR[rt] * NU + tc] = S[rt][2];
                    }
                }
                if (rowl) {
                    const T xn1 = (warm) ? xi[(size_t)(k + 1) * NX + rr] : x0[rr];
                    T xnr = 0;
                    NMPC_UNROLL for (int i = 0; i < NX; i++) xnr = (i == rr) ? xs[i] : xnr;
                    a[ob + rr] = xnr - xn1;
                }
            }
        }
    } else
    for (int k = 0; k < Ns; k++) {
        T xs[NX], us[NU], S[NX];
        NMPC_UNROLL for (int i = 0; i < NX; i++) xs[i] = (warm && k > 0) ? xi[(size_t)k * NX + i] : x0[i];
        NMPC_UNROLL for (int i = 0; i < NU; i++) us[i] = warm ? ui[(size_t)k * NU + i] : T(0);
        NMPC_UNROLL for (int i = 0; i < NX; i++) S[i] = (col < 7 && i == 6 + col) ? T(1) : T(0);
        for (int st = 0; st < c.steps; st++) {
            Jac<T> J1, J2;
            T f1[NX], xm[NX], f2[NX], k1[NX], sm[NX], k2[NX];
            model_f(c, xs, us, f1);
            model_jac(c, xs, us, J1);
            NMPC_UNROLL for (int i = 0; i < NX; i++) xm[i] = xs[i] + T(0.5) * c.h * f1[i];
            model_f(c, xm, us, f2);
            model_jac(c, xm, us, J2);
            NMPC_UNROLL for (int i = 0; i < NX; i++) xs[i] += c.h * f2[i];
            vde_col_rt(c, J1, S, k1, is_u, col - 7);
            NMPC_UNROLL for (int i = 0; i < NX; i++) sm[i] = S[i] + T(0.5) * c.h * k1[i];
            vde_col_rt(c, J2, sm, k2, is_u, col - 7);
            NMPC_UNROLL for (int i = 0; i < NX; i++) S[i] += c.h * k2[i];
        }
        const bool to_lds = smem != nullptr && c.shared;
        if (valid || to_lds) {
            T *a = to_lds ? smem + team * TEAM_LDS : w.tAB + ((size_t)inst * Ns + k) * TAB_ROWS;
            const int oA = to_lds ? L_AD : 0, oB = to_lds ? L_B : 104, ob = to_lds ? L_BV : 156;
            if (r < 7) {
                NMPC_UNROLL for (int i = 0; i < NX; i++) a[oA + i * 8 + r] = S[i];
            } else if (r < 11) {
                NMPC_UNROLL for (int i = 0; i < NX; i++) a[oB + i * NU + (r - 7)] = S[i];
            }
            if (rowl) {
                a[oA + rr * 8 + 7] = 0;
                const T xn1 = (warm) ? xi[(size_t)(k + 1) * NX + rr] : x0[rr];
                T xnr = 0;
                NMPC_UNROLL for (int i = 0; i < NX; i++) xnr = (i == rr) ? xs[i] : xnr;
                a[ob + rr] = xnr - xn1;
            }
        }
    }
    if (valid && rowl) {              // two exec-mask regions for all stages (stage bound: scalar branch)
        NMPC_UNROLL for (int i = 0; i < HC; i++) {
            if (i < N) {
                if (!c.shared || i == 0) NMPC_TST(w.xl, XLR, i * NX + rr, hx[i]);    // shared cold start: stage 0 stands for all
                NMPC_TST(w.qr, QRR, i * QR_ROWS + rr, Wqr * (hx[i] - hyx[i]));
            }
        }
    }
    if (valid && cmpl) {
        NMPC_UNROLL for (int i = 0; i < HC; i++) {
            if (i < N) {
                if (!c.shared) NMPC_TST(w.ul, ULR, i * NU + j, hu[i]);     // shared cold start: u_k = 0, folded in team_ipm
                NMPC_TST(w.qr, QRR, i * QR_ROWS + NX + j, Wrj * (hu[i] - hyu[i]));
            }
        }
    }
    for (int k0 = HC; k0 < N; k0 += CH) {
        T xv[CH], uv[CH], yx[CH], yu[CH];
        NMPC_UNROLL for (int i = 0; i < CH; i++) {
            const int k = (k0 + i < N) ? k0 + i : N - 1;
            xv[i] = warm ? xi[(size_t)k * NX + rr] : x0r;
            uv[i] = warm ? ui[(size_t)k * NU + j] : T(0);
            yx[i] = yr[(size_t)k * NY + rr];
            yu[i] = yr[(size_t)k * NY + NX + j];
        }
        NMPC_UNROLL for (int i = 0; i < CH; i++) {
            const int k = k0 + i;
            if (k < N && valid) {
                if (rowl) {
                    if (!c.shared) NMPC_TST(w.xl, XLR, k * NX + rr, xv[i]);
                    NMPC_TST(w.qr, QRR, k * QR_ROWS + rr, Wqr * (xv[i] - yx[i]));
                }
                if (cmpl) {
                    if (!c.shared) NMPC_TST(w.ul, ULR, k * NU + j, uv[i]);
                    NMPC_TST(w.qr, QRR, k * QR_ROWS + NX + j, Wrj * (uv[i] - yu[i]));
                }
            }
        }
    }
    if (rowl && valid) {
        if (!c.shared) NMPC_TST(w.xl, XLR, N * NX + rr, xN);
        NMPC_TST(w.qr, QRR, N * QR_ROWS + rr, WqNr * (xN - yeN));
    }
}

#endif  // device

}  // namespace nmpc
