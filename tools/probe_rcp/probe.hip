// probe_rcp: accuracy of v_rcp_f64 followed by 0, 1 or 2 Newton steps against the IEEE quotient, over 2^24 random arguments.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/probe_rcp/probe tools/probe_rcp/probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const double *x, double *e, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i], q = 1.0 / v;
    double r = __builtin_amdgcn_rcp(v);
    e[i] = fabs(r - q) / fabs(q);
    r = __builtin_fma(__builtin_fma(-v, r, 1.0), r, r);
    e[n + i] = fabs(r - q) / fabs(q);
    r = __builtin_fma(__builtin_fma(-v, r, 1.0), r, r);
    e[2 * n + i] = fabs(r - q) / fabs(q);
}
int main()
{
    const int n = 1 << 24;
    std::vector<double> h(n), e(3 * n);
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double m = 1.0 + (double)(s >> 11) / 9007199254740992.0;          // [1, 2)
        const int ex = (int)((s >> 3) % 120) - 60;
        h[i] = std::ldexp(m, ex) * ((s & 1) ? 1.0 : -1.0);
    }
    double *dx, *de;
    hipMalloc(&dx, n * 8); hipMalloc(&de, 3 * n * 8);
    hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, de, n);
    hipMemcpy(e.data(), de, 3 * n * 8, hipMemcpyDeviceToHost);
    for (int j = 0; j < 3; j++) {
        double mx = 0, sm = 0;
        for (int i = 0; i < n; i++) { mx = std::fmax(mx, e[(size_t)j * n + i]); sm += e[(size_t)j * n + i]; }
        std::printf("v_rcp_f64 + %d Newton steps: max relative error %.3e (%.2f ulp of 2^-53), mean %.3e\n", j, mx, mx / 1.1102230246251565e-16, sm / n);
    }
    return 0;
}
