// nmpc_as.hip -- k_team_as, the kernel of the default FP64 path (first active-set attempt of every instance and, on the same wave, the
// continuation of the attempts that fail), built with the two internal LLVM options of the Makefile (-amdgpu-mfma-vgpr-form,
// -amdgpu-sched-strategy=iterative-ilp).  nmpc_qp.hip holds the twin built without them, which this build is held bit-equal to.
#include <hip/hip_runtime.h>

#include "nmpc_as_launch.hpp"

using namespace nmpc;

namespace {

// first launch of the default FP64 path: preparation + the first active-set attempt (nmpc_team_as.hpp).
// OCC = waves per SIMD the register allocation allows: 2 (256 registers) pays once the batch supplies two waves
// per SIMD (B >= 8192); below that one wave per SIMD is all there is and the 512-register build has no spills.
template <bool SHARED, bool TRAJ, int OCC, class TI>
__global__ __launch_bounds__(64, OCC) void k_team_as(const Consts<double> *__restrict__ cp, Work<double> w, Inputs<TI> in, Outputs<TI> out,
                                                     TeamWork<double> tw, WorkList wl, int B, int tpw, int lds_stride, int lstg, int lm_off,
                                                     int pass_cap, double *tail_ts, int cont_stride, int cont_lstg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // the constant block is read from device memory (uploaded at create): scalar loads on demand for uniform entries,
    // one vector load for a per-lane entry
    team_as_kernel<SHARED, TRAJ, OCC == 1, OCC == 1 && as_cont_built(SHARED, TRAJ), TI>(*cp, w, in, out, tw, wl, B, tpw, reinterpret_cast<double *>(smem_raw), lds_stride, lstg,
                                                          lm_off, pass_cap, tail_ts, cont_stride, cont_lstg);
}

template <class TI>
int launch_impl(const AsLaunch &a, const Inputs<TI> &in, const Outputs<TI> &out)
{
    const dim3 grid((a.B + a.tpw - 1) / a.tpw), block(64);
#define NMPC_LAUNCH_AS(SH_, TR_, OC_) hipLaunchKernelGGL((k_team_as<SH_, TR_, OC_, TI>), grid, block, a.lds_bytes, a.stream, a.cp, a.w, in, out, a.tw, a.wl, a.B, a.tpw, a.lds_stride, a.lstg, a.lm_off, a.tail.cap, a.tail.ts, a.cont_stride, a.cont_lstg)
    if (a.shared) {
        if (a.occ == 2) { if (a.traj) NMPC_LAUNCH_AS(true, true, 2); else NMPC_LAUNCH_AS(true, false, 2); }
        else { if (a.traj) NMPC_LAUNCH_AS(true, true, 1); else NMPC_LAUNCH_AS(true, false, 1); }
    } else {
        if (a.traj) NMPC_LAUNCH_AS(false, true, 1); else NMPC_LAUNCH_AS(false, false, 1);
    }
#undef NMPC_LAUNCH_AS
    return (int)hipGetLastError();
}

}  // namespace

namespace nmpc {
int launch_team_as(const AsLaunch &a, const Inputs<double> &in, const Outputs<double> &out) { return launch_impl<double>(a, in, out); }
int launch_team_as(const AsLaunch &a, const Inputs<float> &in, const Outputs<float> &out) { return launch_impl<float>(a, in, out); }
}  // namespace nmpc
