// nmpc_as_launch.hpp -- host-side hand-over between the C ABI (nmpc_capi.hip) and the translation unit that holds the
// active-set kernels (nmpc_as.hip).  The two are compiled separately because the active-set kernels are built with
// -mllvm -amdgpu-mfma-vgpr-form (MFMA results in the vector registers the following VALU reads: +9 % on them), a flag
// under which this compiler miscounts the interior-point iterations of the general kernel (results stay right; found by
// the iteration statistics of the plain-IPM parity run), so the general kernels keep the default code generation.
#pragma once

#include <hip/hip_runtime.h>

#include "nmpc_team_as.hpp"

namespace nmpc {

struct AsLaunch {
    const Consts<double> *cp;   // constant block in device memory
    Work<double> w;
    TeamWork<double> tw;
    WorkList wl;
    int B, tpw, lds_stride, lstg, occ;
    int lm_off = 0;             // LDS offset (doubles) of a team's stage cache behind its working arrays
    int kind = 0;               // 0: k_team_as (first attempt), 1: k_team_qp (whole QP, whole batch), 2: k_team_qp_list (work list)
    int nlist = 0;              // workgroups of the work-list launch
    bool shared, traj;
    size_t lds_bytes;
    hipStream_t stream;
};

// enqueue k_team_as<shared, traj, occ, TI> / k_team_qp / k_team_qp_list (kind); returns a hipError_t
int launch_team_as(const AsLaunch &a, const Inputs<double> &in, const Outputs<double> &out);
int launch_team_as(const AsLaunch &a, const Inputs<float> &in, const Outputs<float> &out);

}  // namespace nmpc
