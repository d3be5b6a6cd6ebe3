// Probe of v_mfma_f64_4x4x4_4b_f64 on gfx950: operand/result lane layout and issue cost.
// build: hipcc -O3 --offload-arch=gfx950 -o probe probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_layout(double *out)
{
    const int l = threadIdx.x;
    for (int x = 0; x < 16; x++)
        for (int y = 0; y < 16; y++) {
            const double a = ((l & 15) == x) ? 1.0 : 0.0;
            const double b = ((l & 15) == y) ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            out[(x * 16 + y) * 64 + l] = d;
        }
}

// cross-block check: does block 1's result depend on block 0's operands?
__global__ void k_blocks(double *out)
{
    const int l = threadIdx.x;
    const double a = (l < 16) ? 1.0 : 0.0, b = 1.0;
    out[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
}

template <int CHAINS>
__global__ void k_time(double *out, long long *cyc, int iters)
{
    const int l = threadIdx.x;
    double acc[CHAINS];
    for (int c = 0; c < CHAINS; c++) acc[c] = l * 0.001 + c;
    const double a = 1.0 + l * 1e-3, b = 0.5 - l * 1e-3;
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
    }
    const long long t1 = clock64();
    double s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c];
    out[blockIdx.x * 64 + l] = s;
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CHAINS>
__global__ void k_time_fma(double *out, long long *cyc, int iters)
{
    const int l = threadIdx.x;
    double acc[CHAINS];
    for (int c = 0; c < CHAINS; c++) acc[c] = l * 0.001 + c;
    const double a = 1.0 + l * 1e-9, b = 0.5 - l * 1e-3;
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_fma(a, acc[c], b);
    }
    const long long t1 = clock64();
    double s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c];
    out[blockIdx.x * 64 + l] = s;
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main()
{
    double *d;
    long long *c;
    CK(hipMalloc(&d, 256 * 64 * sizeof(double)));
    CK(hipMalloc(&c, 4096 * sizeof(long long)));
    std::vector<double> h(256 * 64);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, d);
    CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
    printf("layout (block 0): a one at lane x of A and lane y of B -> lanes of D that become 1\n");
    for (int x = 0; x < 16; x++) {
        printf("A lane %2d:", x);
        for (int y = 0; y < 16; y++) {
            int hits = 0, where = -1;
            for (int l = 0; l < 16; l++)
                if (h[(x * 16 + y) * 64 + l] != 0.0) { hits++; where = l; }
            if (hits == 0) printf("  .");
            else if (hits == 1) printf(" %2d", where);
            else printf(" *%d", hits);
        }
        printf("\n");
    }
    bool same = true;
    for (int x = 0; x < 16 && same; x++)
        for (int y = 0; y < 16 && same; y++)
            for (int l = 0; l < 16; l++)
                for (int b = 1; b < 4; b++)
                    if (h[(x * 16 + y) * 64 + l] != h[(x * 16 + y) * 64 + 16 * b + l]) same = false;
    printf("blocks 1..3 follow the same pattern within their own 16 lanes: %s\n", same ? "yes" : "NO");
    hipLaunchKernelGGL(k_blocks, dim3(1), dim3(64), 0, 0, d);
    CK(hipMemcpy(h.data(), d, 64 * 8, hipMemcpyDeviceToHost));
    printf("A nonzero in block 0 only, B all ones: D =");
    for (int l = 0; l < 64; l++) printf(" %g", h[l]);
    printf("\n");

    const int iters = 4096;
    std::vector<long long> hc(4096);
    auto report = [&](const char *name, int chains, int nblk) {
        CK(hipMemcpy(hc.data(), c, nblk * 8, hipMemcpyDeviceToHost));
        double m = 0;
        for (int i = 0; i < nblk; i++) m += hc[i];
        m /= nblk;
        printf("%-28s chains %d, %4d waves: %.2f clock64 ticks per instruction\n", name, chains, nblk, m / ((double)iters * chains));
        return 0;
    };
    for (int nblk : {1, 1024, 4096}) {
        hipLaunchKernelGGL(k_time<1>, dim3(nblk), dim3(64), 0, 0, d, c, iters); CK(hipDeviceSynchronize()); report("mfma_f64_4x4x4 dependent", 1, nblk);
        hipLaunchKernelGGL(k_time<4>, dim3(nblk), dim3(64), 0, 0, d, c, iters); CK(hipDeviceSynchronize()); report("mfma_f64_4x4x4", 4, nblk);
        hipLaunchKernelGGL(k_time<8>, dim3(nblk), dim3(64), 0, 0, d, c, iters); CK(hipDeviceSynchronize()); report("mfma_f64_4x4x4", 8, nblk);
        hipLaunchKernelGGL(k_time_fma<1>, dim3(nblk), dim3(64), 0, 0, d, c, iters); CK(hipDeviceSynchronize()); report("v_fma_f64 dependent", 1, nblk);
        hipLaunchKernelGGL(k_time_fma<8>, dim3(nblk), dim3(64), 0, 0, d, c, iters); CK(hipDeviceSynchronize()); report("v_fma_f64", 8, nblk);
    }
    // wall-clock rate for the chip-filling case
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_time<8>, dim3(4096), dim3(64), 0, 0, d, c, iters);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("mfma 4x4x4 x8 chains, 4096 waves: %.3f ms -> %.1f TFLOP/s\n", ms, 4096.0 * iters * 8 * 512 / (ms * 1e-3) / 1e12);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_time_fma<8>, dim3(4096), dim3(64), 0, 0, d, c, iters);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("v_fma_f64 x8 chains, 4096 waves: %.3f ms -> %.1f TFLOP/s\n", ms, 4096.0 * iters * 8 * 128 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
