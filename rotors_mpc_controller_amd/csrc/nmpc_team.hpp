// nmpc_team.hpp -- what the kernels of the "team" mapping share (gfx950 device code): 16 lanes per MPC instance.
//
// Why teams: with one instance per lane a batch of 4096 is only 64 waves on a chip with 1024 SIMDs and
// each wave streams ~40 KB of private workspace per instance through HBM/L2 -- measured
// memory-latency bound (profiles/).  Here a wave holds 4 instances ("teams"); team b owns
// lanes {16a + 4b + c}, lane (a,c) being element (a,c) of every 4x4 register tile of the sweeps, which run
// on v_mfma_f64_4x4x4_4b_f64 (whose block layout this is).  One wave per workgroup makes every LDS exchange a
// single-wave hand-off.  B = 4096 -> 1024 waves = one per SIMD.
//
// This header holds the workspace layout, the MFMA / reciprocal helpers, the bound-pair algebra of the interior
// point and the tile form of the model Jacobian.  The kernels themselves: nmpc_team_as.hpp (k_team_as, k_team_qp,
// k_team_qp_list, k_team_tail), nmpc_stage.hpp (the factor stage they share), nmpc_block.hpp (block-parallel sweeps).
// (Rounds 1-4 also kept the first team kernel here - k_team_ipm, row-per-lane sweeps, the only FP32-arithmetic
// path - retired in round 5: NMPC_DTYPE_F32 now means FP32 buffers on the FP64 kernels)
//
// Arithmetic short-cuts that change results at rounding level only (tests hold 1e-9 against the oracle): slack
// reciprocals (v_rcp_f64 + 2 Newton steps) replace the IEEE divisions, the pivots use v_rsq_f64 + 2 Newton
// steps, and the tile form sums in MFMA order.
//
// Per-instance scratch in HBM is "array of structures" (a team reads contiguous runs):
//   tLM [inst][stage][TLM_ROWS]: Mbar^T tiles (64) | L^-1 tile (16) | gradient rows of stages with pins | xhat
//   tIV [inst][stage][IV_ROWS] : u | lam_l | lam_u | affine step | step or pin codes | spare | warm start | t_l | t_u
//   tP  [inst][1 + ckpt][256]  : (P_k, p_k), k = 1..ckpt, as left by an active-set pass (16 tiles x 16 lanes of
//                           Pbar).  The factorisation
//                           of stage k depends on the pins of stages >= k only, so the next pass restarts its
//                           backward sweep at the highest stage whose pin set changed - when that lies in the
//                           checkpointed window (saturation sits in the first stages of the horizon) - instead
//                           of at N-1.  The window bounds the store traffic: 1.4 KB per stage and instance
//                           through a 64 B/clk store path cost 7 % of the sweep when every stage was kept.
// The stage matrices come from the per-instance block tAB written by the preparation of nmpc_team_as.hpp.
#pragma once

#include "nmpc_lane.hpp"

namespace nmpc {

constexpr int TEAM = 16;            // lanes per instance
constexpr int TEAMS_PER_WAVE = 4;
constexpr int TLM_ROWS = 240;      // per stage: [52..63] 1 / d | [66..79] xhat of the forward sweep | [80..143] Mbar^T as 4 tiles x 16 lanes |
                                   // [144..159] L^-1 tile | [160..239] stages with pins: (B'Pbar Abar)^T unmasked (64) | (B'PB)^T (16)
constexpr int TLM_MT = 80, TLM_Z = 144, TLM_G = 160;
constexpr int TLM_RINV = 52;       // H_uu = L D L': the four 1 / d_a of a stage (52..63)
// TAB_ROWS (nmpc_lane.hpp): 104 (Ad rows, 8 each) + 52 (B rows) + 13 (b) + pad here; 12 tiles x 16 in the active-set kernel
constexpr int TP_ROWS = 256;        // Riccati checkpoint of a stage: 16 tiles x 16 lanes of Pbar

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

// hard fence for the machine scheduler: nothing is moved across it (used to keep LDS reads batched)
#define NMPC_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double fast_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    y = __builtin_fma(0.5 * y, __builtin_fma(-x * y, y, 1.0), y);
    y = __builtin_fma(0.5 * y, __builtin_fma(-x * y, y, 1.0), y);
    return y;
}

// D = A^T * B + C on 4x4 tiles of four independent blocks (= the four teams of a wave): element (a,c) of
// every operand and of the result sits in lane 16a + 4b + c of block b, so the A operand is read as the
// TRANSPOSE of the tile stored that way (layout and rate probed with tools/probe_mfma: one wave alone
// reaches the full FP64 rate with this instruction, but only half of it with v_fma_f64).
__device__ __forceinline__ double mfma44(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
// (-X)'Y + C: the FP64 MFMAs take a negation per operand in their BLGP field (neg:[1,0,0]) - exact, and one v_xor_b32 per tile saved
__device__ __forceinline__ double mfma44_na(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 1); }

// A pivot of a Riccati stage is valid while 0 < d <= PIVOT_MAX: NaN, not positive, or a magnitude that can no longer be squared in double
// precision (a linearisation about a diverged trajectory: pivot 8.9e269 in fuzz draw 11856) ends the factorisation instead of running
// on into inf - inf.  Same test, same constant: chol_lower of oracle/nmpc_oracle.c.  (How a failed solve is then CLASSED - status 1 or 4 -
// is decided by the inputs, not by which pivot failed how: inputs_not_finite, nmpc_ipm.hpp.)
constexpr double PIVOT_MAX = 1e100;

// padded state order of the tile form: p(3) _ | v(3) _ | q(4) | omega(3) 1   (index 15 is the homogeneous
// coordinate that carries b, p and the gradients); natural index of element e of tile t, -1 for a pad
__device__ __forceinline__ constexpr int nat_of(int t, int e)
{
    return t == 0 ? (e < 3 ? e : -1) : (t == 1 ? (e < 3 ? 3 + e : -1) : (t == 2 ? 6 + e : (e < 3 ? 10 + e : -1)));
}

// Where the interior point's iterate of input a lives in a stage's row of tIV: u at a; the PAIRS (lam_l, lam_u) at IVP_L + 2a and (t_l, t_u)
// at IVP_T + 2a, 16-byte aligned, so that a sweep fetches each pair with one global_load_dwordx4 (until round 5 the four values sat 32 bytes
// apart: four loads - carrying the slacks had made the interior-point kernel 8 % slower than round 4, mostly through its load count).
constexpr int IVP_L = 4, IVP_T = 36;
struct D2 { double x, y; };
__device__ __forceinline__ D2 ld2(const double *p)
{
    const double2 v = *reinterpret_cast<const double2 *>(p);
    return D2{v.x, v.y};
}
__device__ __forceinline__ void st2(double *p, double x, double y) { *reinterpret_cast<double2 *>(p) = make_double2(x, y); }

// One bound pair (lower, upper) of one input of the interior point's iterate: the slacks t_l, t_u are ITERATES of their own (stored with
// the iterate, t <- t + alpha dt: HPIPM's form, oracle ocpqp_ipm) - never re-formed as u - lo, which cannot resolve the 1e-14 the central
// path asks of an active bound's slack at mu = 1e-11.  For an input step d: dt_l = d, dt_u = -d.  (HPIPM also feeds the residuals of the
// bound equations, u - lo - t_l and hi - u - t_u, back into the right-hand side; here the start is feasible and t follows u exactly in exact
// arithmetic, so they hold rounding of size ulp(u) - the oracle has them behind orc_config.qp_bound_res and tests/test_oracle_qp.py shows
// iteration counts and commands unchanged with them; the kernels do not pay the ~6 FP64 operations per pair and sweep.)
template <class T>
struct Pair {
    T tl, tu, itl, itu, kl, ku;
    __device__ __forceinline__ Pair(T ll, T lu, T tl_, T tu_)
    {
        tl = tl_; tu = tu_;
        itl = fast_rcp(tl); itu = fast_rcp(tu);
        kl = ll * itl; ku = lu * itu;
    }
};

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

#endif  // device

}  // namespace nmpc
