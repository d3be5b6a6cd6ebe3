// nmpc_qpf.hip -- k_team_qp (the whole interior-point QP of every instance of the batch: what qp_polish = 0 runs), k_team_qp_list (the
// work list of the default path) and k_team_tail (the long-horizon tail) built with -mllvm -amdgpu-mfma-vgpr-form, like k_team_as in nmpc_as.hip: the interior point's factor stage is the shared one (nmpc_stage.hpp) and
// pays the same accumulation-register moves in the default code generation (15.7 % of the instructions a wave issues, tools/emu/instr_mix.py).
// The source is nmpc_qp.hip, included with everything but that kernel switched off.  What stands behind the flag here: the GPU test
// test_flag_build_of_the_interior_point_kernel_is_bit_equal_to_the_default_codegen_build (NMPC_QP_NOFLAG=1 selects nmpc_qp.hip's build), and
// the CPU emulation of this very instantiation on the launch that faulted in round 3 (profiles/r04_fault_emulation.txt: clean).
#define NMPC_QP_WHOLE_BATCH_ONLY 1
#define NMPC_QP_EXPORT launch_team_qp_flag
#include "nmpc_qp.hip"
