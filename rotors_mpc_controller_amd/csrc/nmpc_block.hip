// nmpc_block.hip -- kernels of the parallel-in-time Riccati factorisation (nmpc_block.hpp), default code generation.
#include <hip/hip_runtime.h>

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
    if (phase == 1 && blk < g.J - 1) block_sweep<true, TI>(*cp, g, in, blk, smem);
    else block_sweep<false, TI>(*cp, g, in, blk, smem);
}

__global__ __launch_bounds__(64, 1) void k_block_scan(BlockWork g)
{
    __shared__ __attribute__((aligned(16))) double smem[4 * BLK_LDS];
    block_scan(g, smem);
}

template <class TI>
int launch_impl(const BlockLaunch &a, const Inputs<TI> &in)
{
    const BlockWork &g = a.g;
    const dim3 block(64);
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
int launch_block_factor(const BlockLaunch &a, const Inputs<double> &in) { return launch_impl<double>(a, in); }
int launch_block_factor(const BlockLaunch &a, const Inputs<float> &in) { return launch_impl<float>(a, in); }
}  // namespace nmpc
