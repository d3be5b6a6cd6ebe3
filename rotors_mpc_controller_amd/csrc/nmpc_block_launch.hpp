// nmpc_block_launch.hpp -- host-side hand-over between the C ABI (nmpc_capi.hip) and nmpc_block.hip (parallel-in-time Riccati
// factorisation, nmpc_block.hpp).
#pragma once

#include <hip/hip_runtime.h>

#include "nmpc_block.hpp"

namespace nmpc {

struct BlockLaunch {
    const Consts<double> *cp;   // constant block in device memory
    BlockWork g;
    hipStream_t stream;
    hipEvent_t ev[4];           // recorded around the three launches when non-null: [0] start, [1] after launch 1, [2] after the scan, [3] end
    bool timing;
    int tail_grid = 0;          // > 0: tail mode (work list, factors into the solver's rows) with this many workgroups per block
};

// enqueue the factorisation (J = 1: one ordinary sweep per instance, the sequential form in the same code); returns a hipError_t
int launch_block_factor(const BlockLaunch &a, const Inputs<double> &in);
int launch_block_factor(const BlockLaunch &a, const Inputs<float> &in);
// ... the same kernels from nmpc_blockf.hip (built with -amdgpu-mfma-vgpr-form)
int launch_block_factor_flag(const BlockLaunch &a, const Inputs<double> &in);
int launch_block_factor_flag(const BlockLaunch &a, const Inputs<float> &in);

}  // namespace nmpc
