// Micro-benchmark: sustained rate of the f32-input MFMA forms on gfx950, with the accumulator
// pattern k_tile_mfma uses (4 independent 16x16 accumulators, same A operand).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, const float *in, int iters)
{
    float a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    float s = 0;
    if (MODE == 0) { // 16x16x4, 4 accumulators
        f32x4 acc[4];
        for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g], 0, 0, 0);
            }
        }
        for (int g = 0; g < 4; ++g) s += acc[g][0] + acc[g][1] + acc[g][2] + acc[g][3];
    } else if (MODE == 1) { // 32x32x2, 2 accumulators
        f32x16 acc[2];
        for (int g = 0; g < 2; ++g) for (int i = 0; i < 16; ++i) acc[g][i] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int g = 0; g < 2; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g], 0, 0, 0);
            }
        }
        for (int g = 0; g < 2; ++g) for (int i = 0; i < 16; ++i) s += acc[g][i];
    } else { // 16x16x4 with 8 VALU instructions interleaved per 4 MFMAs (address-math stand-in)
        f32x4 acc[4];
        for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0, 0, 0, 0};
        int e = threadIdx.x, off = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g], 0, 0, 0);
                e += 4; off += 4;
                if (e >= iters) { off += 2; e -= 160; }
                asm volatile("" : "+v"(e), "+v"(off));
            }
        }
        for (int g = 0; g < 4; ++g) s += acc[g][0] + acc[g][1] + acc[g][2] + acc[g][3];
        s += off;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> int run(const char *name, int waves_per_simd, double flop_per_mfma, int mfma_per_iter)
{
    const int iters = 2000;
    const int blocks = 256 * waves_per_simd;
    float *out, *in;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHECK(hipMalloc(&in, 1024));
    std::vector<float> h(256); for (int i = 0; i < 256; ++i) h[i] = getenv("UB_RANDOM") ? (float)((i * 2654435761u) >> 8) / 16777216.f - 0.5f : 1e-3f;
    CHECK(hipMemcpy(in, h.data(), 1024, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, in, 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)iters * mfma_per_iter * flop_per_mfma * 4 * blocks;
    printf("%-12s waves/SIMD %d : %8.3f ms  %7.1f TFLOP/s\n", name, waves_per_simd, ms, flops / ms / 1e9);
    CHECK(hipFree(out)); CHECK(hipFree(in));
    return 0;
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("16x16x4 x4", w, 2.0 * 16 * 16 * 4, 16);
        run<1>("32x32x2 x2", w, 2.0 * 32 * 32 * 2, 8);
        run<2>("16x16x4+valu", w, 2.0 * 16 * 16 * 4, 16);
    }
    return 0;
}
