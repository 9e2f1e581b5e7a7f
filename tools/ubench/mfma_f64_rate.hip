// mfma_f64_rate — sustained rate of v_mfma_f64_16x16x4_f64 (2048 flop per wave-instruction): W waves per SIMD, 4
// independent accumulators per wave.   Build: hipcc --offload-arch=gfx950 -O3 mfma_f64_rate.hip -o mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int NACC>
__global__ void k(double *out, const double *in, int iters)
{
    f64x4 acc[NACC];
    const double a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    for (int i = 0; i < NACC; ++i) acc[i] = (f64x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 1.2345) out[threadIdx.x] = s;
}
int main()
{
    double *in, *out;
    CHECK(hipMalloc(&in, 4096)); CHECK(hipMalloc(&out, 8192)); CHECK(hipMemset(in, 0, 4096));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int nacc : {1, 2, 4})
        for (int W : {1, 2, 4}) {
            const dim3 grid(256 * 8), block(64 * 4 * W > 1024 ? 1024 : 64 * 4 * W); // plenty of workgroups: every CU busy
            auto launch = [&]() {
                if (nacc == 1) hipLaunchKernelGGL(k<1>, grid, block, 0, nullptr, out, in, iters);
                else if (nacc == 2) hipLaunchKernelGGL(k<2>, grid, block, 0, nullptr, out, in, iters);
                else hipLaunchKernelGGL(k<4>, grid, block, 0, nullptr, out, in, iters);
            };
            launch(); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, nullptr)); launch(); CHECK(hipEventRecord(e1, nullptr)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double mfmas = (double)grid.x * (block.x / 64) * iters * 8 * nacc;
            printf("accumulators %d, block %4d threads: %8.1f us  %6.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", nacc, block.x, ms * 1e3,
                   mfmas * 2048 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (mfmas / 1024));
        }
    return 0;
}
