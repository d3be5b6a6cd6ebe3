// nmpc_as_launch.hpp -- host-side hand-over between the C ABI (nmpc_capi.hip) and the translation units that hold the kernels of
// nmpc_team_as.hpp: nmpc_as.hip (k_team_as) and nmpc_qpf.hip (k_team_qp, k_team_qp_list, k_team_tail) - what runs, built with the internal
// LLVM options of the Makefile - and nmpc_qp.hip (all four, default code generation: the twins the flag builds are held bit-equal to, and
// what sim_num_steps > 2 runs).
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
    int kind = 0;               // 0: k_team_as (first attempt), 1: k_team_qp (whole QP, whole batch), 2: k_team_qp_list (work list),
                                // 3: k_team_tail (one step of the block-parallel tail), 4: reset of a work list
    TailCtx tail;               // kind 3
    int nlist = 0;              // workgroups of the work-list launch
    int cont_stride = 0, cont_lstg = 0;   // kind 0, one-wave builds: > 0 = failed first attempts continue on their own wave (team stride and cached
                                // stages of the MODE 2 carve: IP_LM_ROWS per stage); 0 = they go to the work list
    bool shared, traj;
    size_t lds_bytes;
    hipStream_t stream;
};

// enqueue k_team_as<shared, traj, occ, TI> of nmpc_as.hip (kind 0 only); returns a hipError_t
int launch_team_as(const AsLaunch &a, const Inputs<double> &in, const Outputs<double> &out);
int launch_team_as(const AsLaunch &a, const Inputs<float> &in, const Outputs<float> &out);
// enqueue a kernel of nmpc_qp.hip: k_team_as (kind 0), k_team_qp (1), k_team_qp_list (2)
int launch_team_qp(const AsLaunch &a, const Inputs<double> &in, const Outputs<double> &out);
int launch_team_qp(const AsLaunch &a, const Inputs<float> &in, const Outputs<float> &out);
// kinds 1-3 of nmpc_qpf.hip: the same source built with the Makefile's internal LLVM options
int launch_team_qp_flag(const AsLaunch &a, const Inputs<double> &in, const Outputs<double> &out);
int launch_team_qp_flag(const AsLaunch &a, const Inputs<float> &in, const Outputs<float> &out);

}  // namespace nmpc
