// nmpc_block.hip -- kernels of the parallel-in-time Riccati factorisation (nmpc_block.hpp), default code generation.
// (nmpc_blockf.hip includes this file to build the same kernels with -amdgpu-mfma-vgpr-form under another launcher name.)
#include <hip/hip_runtime.h>
#ifndef NMPC_BLOCK_EXPORT
#define NMPC_BLOCK_EXPORT launch_block_factor
#endif

#include "nmpc_block_launch.hpp"

using namespace nmpc;

namespace {

// launch 1 (blockIdx.y = block: 0 .. J-2 aggregate, J-1 the ordinary sweep from the terminal cost) and launch 3
// (blockIdx.y = block 0 .. J-2, from the boundary values of the scan).  Teams of a wave share the block index: the branch is uniform.
template <class TI>
__global__ __launch_bounds__(64, 1) void k_block_sweep(const Consts<double> *__restrict__ cp, BlockWork g, Inputs<TI> in, int phase)
{
    __shared__ __attribute__((aligned(16))) double smem[4 * BLK_LDS];
    const int blk = blockIdx.y;
    const int team = (threadIdx.x >> 2) & 3;
    int inst = blockIdx.x * 4 + team;
    const bool valid = inst < g.B;
    if (!valid) inst = g.B - 1;                       // idle teams read the last instance and write the spare row
    if (phase == 1 && blk < g.J - 1) block_sweep<true, TI>(*cp, g, in, blk, smem, inst, valid);
    else block_sweep<false, TI>(*cp, g, in, blk, smem, inst, valid);
}

__global__ __launch_bounds__(64, 1) void k_block_scan(BlockWork g)
{
    __shared__ __attribute__((aligned(16))) double smem[4 * BLK_LDS];
    const int team = (threadIdx.x >> 2) & 3;
    const int inst = blockIdx.x * 4 + team;
    block_scan(g, smem, inst < g.B ? inst : g.B - 1, inst < g.B);
}

// The same three launches for the work list of a long-horizon solve (tail mode): a fixed grid strides over the list; what a team
// factorises - pins of an active-set pass or barrier terms of an interior-point iteration - is its instance's tail state
__device__ __forceinline__ bool tail_pick(const BlockWork &g, int base, int n, int &inst)
{
    const int team = (threadIdx.x >> 2) & 3;
    const int e = base + team;
    int i = g.list[e < n ? e : base];
    const int st = (int)g.ts[(size_t)i * TS_ROWS];
    const bool act = e < n && (st == TS_IPM || st == TS_AS);
    inst = i;
    return act;
}
template <class TI>
__global__ __launch_bounds__(64, 1) void k_block_sweep_tail(const Consts<double> *__restrict__ cp, BlockWork g, Inputs<TI> in, int phase)
{
    __shared__ __attribute__((aligned(16))) double smem[4 * BLK_LDS];
    const int blk = blockIdx.y;
    const int n = *g.count;
    for (int base = blockIdx.x * 4; base < n; base += gridDim.x * 4) {
        int inst;
        const bool act = tail_pick(g, base, n, inst);
        if (__ballot(act) == 0) continue;
        if (phase == 3 && blk == g.J - 1) {
            // slice y = J - 1 of launch 3: the scan's forward walk (state at the start of every block), beside the final sweeps
            block_scan_forward(g, inst, act);
            continue;
        }
        if (phase == 1 && blk < g.J - 1) {
            // aggregate of a block: unchanged since the last pass of this attempt if no pin code of the block changed
            const bool keep = act && g.frec && g.ts[(size_t)inst * TS_ROWS + 10] != 0.0 &&
                              g.frec[((size_t)inst * g.J + blk) * FR_ROWS] == 0.0;
            if (__ballot(act && !keep) == 0) continue;
            block_sweep<true, TI, true>(*cp, g, in, blk, smem, inst, act && !keep);
        }
        else block_sweep<false, TI, true>(*cp, g, in, blk, smem, inst, act);
        __syncthreads();
    }
}
__global__ __launch_bounds__(64, 1) void k_block_scan_tail(BlockWork g)
{
    __shared__ __attribute__((aligned(16))) double smem[4 * BLK_LDS];
    const int n = *g.count;
    if (blockIdx.x == 0 && threadIdx.x == 0 && g.reset_count) *g.reset_count = 0;
    for (int base = blockIdx.x * 4; base < n; base += gridDim.x * 4) {
        int inst;
        const bool act = tail_pick(g, base, n, inst);
        if (__ballot(act) == 0) continue;
        block_scan(g, smem, inst, act);
        __syncthreads();
    }
}

template <class TI>
int launch_impl(const BlockLaunch &a, const Inputs<TI> &in)
{
    const BlockWork &g = a.g;
    const dim3 block(64);
    if (a.tail_grid > 0) {
        const unsigned gx = (unsigned)a.tail_grid;
        hipLaunchKernelGGL((k_block_sweep_tail<TI>), dim3(gx, (unsigned)g.J), block, 0, a.stream, a.cp, g, in, 1);
        hipLaunchKernelGGL(k_block_scan_tail, dim3(gx), block, 0, a.stream, g);
        // (blocks 0 .. J-2: final sweeps; with fwd_in_sweep one more slice, y = J - 1: the forward walk of the boundary scan)
        hipLaunchKernelGGL((k_block_sweep_tail<TI>), dim3(gx, (unsigned)(g.J - 1 + ((g.gbuf && g.fwd_in_sweep) ? 1 : 0))), block, 0, a.stream, a.cp, g, in, 3);
        return (int)hipGetLastError();
    }
    const unsigned gx = (unsigned)((g.B + 3) / 4);
    if (a.timing) (void)hipEventRecord(a.ev[0], a.stream);
    hipLaunchKernelGGL((k_block_sweep<TI>), dim3(gx, (unsigned)g.J), block, 0, a.stream, a.cp, g, in, 1);
    if (a.timing) (void)hipEventRecord(a.ev[1], a.stream);
    if (g.J > 1) {
        hipLaunchKernelGGL(k_block_scan, dim3(gx), block, 0, a.stream, g);
        if (a.timing) (void)hipEventRecord(a.ev[2], a.stream);
        hipLaunchKernelGGL((k_block_sweep<TI>), dim3(gx, (unsigned)(g.J - 1)), block, 0, a.stream, a.cp, g, in, 3);
    } else if (a.timing) {
        (void)hipEventRecord(a.ev[2], a.stream);
    }
    if (a.timing) (void)hipEventRecord(a.ev[3], a.stream);
    return (int)hipGetLastError();
}

}  // namespace

namespace nmpc {
int NMPC_BLOCK_EXPORT(const BlockLaunch &a, const Inputs<double> &in) { return launch_impl<double>(a, in); }
int NMPC_BLOCK_EXPORT(const BlockLaunch &a, const Inputs<float> &in) { return launch_impl<float>(a, in); }
}  // namespace nmpc
