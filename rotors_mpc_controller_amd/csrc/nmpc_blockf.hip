// nmpc_blockf.hip -- the kernels of nmpc_block.hip built with -mllvm -amdgpu-mfma-vgpr-form (as nmpc_as.hip / nmpc_qpf.hip: the factor stage
// is the shared one of nmpc_stage.hpp, and the accumulation-register moves of the default code generation are ~15 % of what a wave issues).
// What the long-horizon tail runs (N >= 160); NMPC_BLOCK_NOFLAG=1 selects nmpc_block.hip's build, and
// tests/test_gpu_block.py::test_flag_build_of_the_block_kernels_is_bit_equal_to_the_default_codegen_build holds the two bit-equal.
#define NMPC_BLOCK_EXPORT launch_block_factor_flag
#include "nmpc_block.hip"
