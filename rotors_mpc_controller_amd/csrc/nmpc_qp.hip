// nmpc_qp.hip -- the kernels of nmpc_team_as.hpp with the compiler's DEFAULT code generation: k_team_as (first launch), k_team_qp (the whole QP,
// whole batch), k_team_qp_list (work-list continuation), k_team_tail (long-horizon tail).
//
// What runs by default is the -mllvm -amdgpu-mfma-vgpr-form build of the same sources (nmpc_as.hip: k_team_as; nmpc_qpf.hip, which includes
// this file: the other three) - an INTERNAL LLVM option, so every flag build is held bit-equal to this file's build of the same kernel on
// the GPU (NMPC_AS_NOFLAG / NMPC_QP_NOFLAG select this file; tests/test_gpu_parity.py) and is executed instruction by instruction on the CPU
// by tools/emu with every address checked (tests/test_isa_emulation.py).  This file's builds are also what sim_num_steps > 2 runs.
// (Round 3 recorded a GPU memory fault on sim_num_steps = 4 inputs of k_team_qp<per-stage, trajectories> and wrote that the flag "miscompiles" it.
// Round 4 emulated that launch on the flag build - all 256 workgroups, 32 M instructions, no access outside a buffer, the oracle's
// answers - and withdrew the claim.  The cause was NOT recovered: no committed state reproduces the fault (the binary that faulted came from a
// working tree that was not kept).  The flag builds stay limited to sim_num_steps <= 2 (launch_qp_kind); docs/history has the examination.)
#include <hip/hip_runtime.h>

#include "nmpc_as_launch.hpp"

using namespace nmpc;

// nmpc_qpf.hip includes this file with NMPC_QP_WHOLE_BATCH_ONLY (everything but k_team_as); the exported launcher then has another name
#ifndef NMPC_QP_EXPORT
#define NMPC_QP_EXPORT launch_team_qp
#endif

namespace {

#ifndef NMPC_QP_WHOLE_BATCH_ONLY
// first launch of the default FP64 path: preparation + the first active-set attempt (nmpc_team_as.hpp).
// OCC = waves per SIMD the register allocation allows: 2 (256 registers) pays once the batch supplies two waves
// per SIMD (B >= 8192); below that one wave per SIMD is all there is and the 512-register build has no spills.
// OCC = 3: the 256-register budget of OCC = 2 WITH the LDS stage cache (40 KB of LDS per wave keep it at one wave per SIMD anyway).  With
// at most 256 registers the compiler's own, supported heuristic selects the VGPR form of the MFMAs (no accumulation registers exist for
// the kernel) - the code shape the internal option -amdgpu-mfma-vgpr-form forces on nmpc_as.hip's 512-register build - and spills what
// does not fit to scratch instead of accumulation registers.  NMPC_AS_BUILD=v256 selects it (shared linearisation only).
template <bool SHARED, bool TRAJ, int OCC, class TI>
__global__ __launch_bounds__(64, OCC == 1 ? 1 : 2) void k_team_as(const Consts<double> *__restrict__ cp, Work<double> w, Inputs<TI> in, Outputs<TI> out,
                                                     TeamWork<double> tw, WorkList wl, int B, int tpw, int lds_stride, int lstg, int lm_off,
                                                     int pass_cap, double *tail_ts, int cont_stride, int cont_lstg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    team_as_kernel<SHARED, TRAJ, OCC != 2, OCC == 1 && as_cont_built(SHARED, TRAJ), TI>(*cp, w, in, out, tw, wl, B, tpw, reinterpret_cast<double *>(smem_raw), lds_stride, lstg,
                                                          lm_off, pass_cap, tail_ts, cont_stride, cont_lstg);
}

#endif  // !NMPC_QP_WHOLE_BATCH_ONLY (k_team_as)

// The whole QP of every instance of the batch in one launch (team_as MODE 1): interior-point iterations in the tile form,
// active-set attempts in between when qp_polish is on.  What qp_polish = 0 runs, and NMPC_TEAM_SPLIT=0.  One wave per SIMD.
template <bool SHARED, bool TRAJ, class TI>
__global__ __launch_bounds__(64, 1) void k_team_qp(const Consts<double> *__restrict__ cp, Work<double> w, Inputs<TI> in, Outputs<TI> out,
                                                   TeamWork<double> tw, WorkList wl, int B, int tpw, int lds_stride, int lstg, int lm_off)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    team_as<SHARED, TRAJ, true, TI, 1>(*cp, w, in, out, tw, wl, B, tpw, reinterpret_cast<double *>(smem_raw), lds_stride, lstg, lm_off);
}

// Second launch of the default FP64 path (team_as MODE 2): the instances the active-set kernel appended to the work list -
// usually none - continue where its first attempt ended: interior-point iterations, further attempts from the iterate's
// active-set guess.  A fixed small grid strides over the list; the last workgroup to finish resets it for the next solve.
template <bool SHARED, bool TRAJ, class TI>
__global__ __launch_bounds__(64, 1) void k_team_qp_list(const Consts<double> *__restrict__ cp, Work<double> w, Inputs<TI> in, Outputs<TI> out,
                                                        TeamWork<double> tw, WorkList wl, int B, int lds_stride, int lstg, int lm_off)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int n = *wl.count;
    // the usual case on the headline workload: nothing was handed over.  Every workgroup reads the same 0 (nobody appends while this
    // kernel runs), the list needs no reset, and the launch ends on one scalar load instead of an atomic round trip per workgroup
    if (n == 0) return;
    const int team = (threadIdx.x >> 2) & 3;
    for (int base = blockIdx.x * 4; base < n; base += gridDim.x * 4) {
        const int e = base + team;
        const int inst = e < n ? wl.list[e] : -1;
        team_as<SHARED, TRAJ, true, TI, 2>(*cp, w, in, out, tw, wl, B, 4, reinterpret_cast<double *>(smem_raw), lds_stride, lstg, lm_off, inst);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int t = atomicAdd(wl.done, 1);          // every workgroup has read the count before it arrives here
        if (t == (int)gridDim.x - 1) { *wl.count = 0; *wl.done = 0; }
    }
}

// One step of the block-parallel tail of a long-horizon solve (team_as MODE 3): the work-list instances whose tail state asks for
// this phase; the factorisation of the step was done by the launches of nmpc_block.hip.  The list is NOT reset here.
template <bool SHARED, bool TRAJ, class TI>
__global__ __launch_bounds__(64, 1) void k_team_tail(const Consts<double> *__restrict__ cp, Work<double> w, Inputs<TI> in, Outputs<TI> out,
                                                     TeamWork<double> tw, WorkList wl, int B, int lds_stride, int lm_off, TailCtx tcx)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int n = *wl.count;
    const int team = (threadIdx.x >> 2) & 3;
    for (int base = blockIdx.x * 4; base < n; base += gridDim.x * 4) {
        const int e = base + team;
        const int inst = e < n ? wl.list[e] : -1;
        const int st = inst >= 0 ? (int)tcx.ts[(size_t)inst * TS_ROWS] : (int)TS_NONE;
        const bool act = inst >= 0 && (tcx.phase == 0 || (tcx.phase == 1 && st == TS_IPM) || ((tcx.phase == 2 || tcx.phase == 3) && st == TS_AS));
        // an instance that is in the tail but not part of this phase stays on the (compacted) list of the next step
        if (tcx.phase != 3 && inst >= 0 && !act && (st == TS_IPM || st == TS_AS) && (threadIdx.x & 0x33) == 0) {
            const int slot = atomicAdd(tcx.nx_count, 1);
            tcx.nx_list[slot] = inst;
        }
        if (__ballot(act) == 0) continue;
        TailCtx tc2 = tcx;
        tc2.blk = blockIdx.y;                  // (phase 3: one block of the horizon per team; the grid's y dimension is 1 otherwise)
        team_as<SHARED, TRAJ, true, TI, 3>(*cp, w, in, out, tw, wl, B, 4, reinterpret_cast<double *>(smem_raw), lds_stride, 0, lm_off,
                                           act ? inst : -1, tc2);
        __syncthreads();
    }
}

#ifndef NMPC_QP_WHOLE_BATCH_ONLY
__global__ void k_list_reset(WorkList wl, int *second_count) { *wl.count = 0; *wl.done = 0; if (second_count) *second_count = 0; }
#endif

template <class TI>
int launch_impl(const AsLaunch &a, const Inputs<TI> &in, const Outputs<TI> &out)
{
#ifndef NMPC_QP_WHOLE_BATCH_ONLY
    if (a.kind == 4) {
        hipLaunchKernelGGL(k_list_reset, dim3(1), dim3(1), 0, a.stream, a.wl, a.tail.nx_count);
        return (int)hipGetLastError();
    }
#endif
    if (a.kind == 3) {
        const dim3 grid(a.nlist, a.tail.phase == 3 ? a.tail.J : 1), block(64);
#define NMPC_LAUNCH_TL(SH_, TR_) hipLaunchKernelGGL((k_team_tail<SH_, TR_, TI>), grid, block, a.lds_bytes, a.stream, a.cp, a.w, in, out, a.tw, a.wl, a.B, a.lds_stride, a.lm_off, a.tail)
        if (a.shared) { if (a.traj) NMPC_LAUNCH_TL(true, true); else NMPC_LAUNCH_TL(true, false); }
        else { if (a.traj) NMPC_LAUNCH_TL(false, true); else NMPC_LAUNCH_TL(false, false); }
#undef NMPC_LAUNCH_TL
        return (int)hipGetLastError();
    }
    if (a.kind == 1) {
        const dim3 grid((a.B + a.tpw - 1) / a.tpw), block(64);
#define NMPC_LAUNCH_QP(SH_, TR_) hipLaunchKernelGGL((k_team_qp<SH_, TR_, TI>), grid, block, a.lds_bytes, a.stream, a.cp, a.w, in, out, a.tw, a.wl, a.B, a.tpw, a.lds_stride, a.lstg, a.lm_off)
        if (a.shared) { if (a.traj) NMPC_LAUNCH_QP(true, true); else NMPC_LAUNCH_QP(true, false); }
        else { if (a.traj) NMPC_LAUNCH_QP(false, true); else NMPC_LAUNCH_QP(false, false); }
#undef NMPC_LAUNCH_QP
        return (int)hipGetLastError();
    }
    if (a.kind == 2) {
        const dim3 grid(a.nlist), block(64);
#define NMPC_LAUNCH_QL(SH_, TR_) hipLaunchKernelGGL((k_team_qp_list<SH_, TR_, TI>), grid, block, a.lds_bytes, a.stream, a.cp, a.w, in, out, a.tw, a.wl, a.B, a.lds_stride, a.lstg, a.lm_off)
        if (a.shared) { if (a.traj) NMPC_LAUNCH_QL(true, true); else NMPC_LAUNCH_QL(true, false); }
        else { if (a.traj) NMPC_LAUNCH_QL(false, true); else NMPC_LAUNCH_QL(false, false); }
#undef NMPC_LAUNCH_QL
        return (int)hipGetLastError();
    }
#ifdef NMPC_QP_WHOLE_BATCH_ONLY
    return (int)hipErrorInvalidValue;
#else
    const dim3 grid((a.B + a.tpw - 1) / a.tpw), block(64);
#define NMPC_LAUNCH_AS(SH_, TR_, OC_) hipLaunchKernelGGL((k_team_as<SH_, TR_, OC_, TI>), grid, block, a.lds_bytes, a.stream, a.cp, a.w, in, out, a.tw, a.wl, a.B, a.tpw, a.lds_stride, a.lstg, a.lm_off, a.tail.cap, a.tail.ts, a.cont_stride, a.cont_lstg)
    if (a.shared) {
        if (a.occ == 2) { if (a.traj) NMPC_LAUNCH_AS(true, true, 2); else NMPC_LAUNCH_AS(true, false, 2); }
        else if (a.occ == 3) { if (a.traj) NMPC_LAUNCH_AS(true, true, 3); else NMPC_LAUNCH_AS(true, false, 3); }
        else { if (a.traj) NMPC_LAUNCH_AS(true, true, 1); else NMPC_LAUNCH_AS(true, false, 1); }
    } else {
        if (a.traj) NMPC_LAUNCH_AS(false, true, 1); else NMPC_LAUNCH_AS(false, false, 1);
    }
#undef NMPC_LAUNCH_AS
    return (int)hipGetLastError();
#endif
}

}  // namespace

namespace nmpc {
int NMPC_QP_EXPORT(const AsLaunch &a, const Inputs<double> &in, const Outputs<double> &out) { return launch_impl<double>(a, in, out); }
int NMPC_QP_EXPORT(const AsLaunch &a, const Inputs<float> &in, const Outputs<float> &out) { return launch_impl<float>(a, in, out); }
}  // namespace nmpc
