// Micro-benchmark: which part of k_tile_mfma_p's loop structure costs matrix-pipe time?
//  0: 16 MFMAs per iteration, in-place accumulators (baseline)
//  1: + 4 ds_read_b128 per iteration feeding the B operands (data-dependent, waited)
//  2: accumulators written to a DIFFERENT register range than they are read from (ping-pong),
//     non-overlapping
//  3: as 1, plus one global_load_dwordx4 prefetch and the register copy at the loop end
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(640) k(float *out, const float4 *tab, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float xs[];
    for (int i = threadIdx.x; i < 12000; i += blockDim.x) xs[i] = 1e-3f * (i & 7);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 acc[4], acc2[4];
    for (int g = 0; g < 4; ++g) { acc[g] = (f32x4){0, 0, 0, 0}; acc2[g] = acc[g]; }
    float4 ac = tab[lane], an = ac;
    const float *px = xs + (lane & 15) * 44 + (lane >> 4) * 3008 / 4;
    int fo = 0;
    for (int it = 0; it < iters; ++it) {
        float4 b0, b1, b2, b3;
        if (MODE == 3) { an = tab[((it + 1) & 15) * 64 + lane]; __builtin_amdgcn_sched_barrier(0); }
        if (MODE == 1 || MODE == 3) {
            const float *p = px + fo;
            b0 = *(const float4 *)__builtin_assume_aligned(p, 16);
            b1 = *(const float4 *)__builtin_assume_aligned(p + 16 * 44, 16);
            b2 = *(const float4 *)__builtin_assume_aligned(p + 32 * 44, 16);
            b3 = *(const float4 *)__builtin_assume_aligned(p + 48 * 44, 16);
            fo = (fo + 4) & 127;
        } else { b0 = b1 = b2 = b3 = ac; }
        if (MODE == 2) {
#define M4(D, S, AV, C) \
    D[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV, b0.C, S[0], 0, 0, 0); D[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV, b1.C, S[1], 0, 0, 0); \
    D[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV, b2.C, S[2], 0, 0, 0); D[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV, b3.C, S[3], 0, 0, 0);
            M4(acc2, acc, ac.x, x) asm volatile("" : "+v"(acc2[0]), "+v"(acc2[1]), "+v"(acc2[2]), "+v"(acc2[3]));
            M4(acc, acc2, ac.y, y) asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
            M4(acc2, acc, ac.z, z) asm volatile("" : "+v"(acc2[0]), "+v"(acc2[1]), "+v"(acc2[2]), "+v"(acc2[3]));
            M4(acc, acc2, ac.w, w) asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
        } else {
            M4(acc, acc, ac.x, x) M4(acc, acc, ac.y, y) M4(acc, acc, ac.z, z) M4(acc, acc, ac.w, w)
        }
        if (MODE == 3) ac = an;
    }
    float s = 0;
    for (int g = 0; g < 4; ++g) s += acc[g][0] + acc[g][1] + acc[g][2] + acc[g][3] + acc2[g][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> int run(const char *name, int wgs_per_cu)
{
    const int iters = 1200;
    const int blocks = 256 * wgs_per_cu;
    float *out; float4 *tab;
    CHECK(hipMalloc(&out, (size_t)blocks * 640 * 4));
    CHECK(hipMalloc(&tab, 16 * 64 * 16));
    std::vector<float> h(16 * 64 * 4, 1e-3f);
    CHECK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(640), 48128, 0, out, tab, 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(640), 48128, 0, out, tab, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)iters * 16 * 2048.0 * 10 * blocks;
    printf("mode %d %-34s WG(10 waves)/CU %d : %8.3f ms  %7.1f TFLOP/s\n", MODE, name, wgs_per_cu, ms, flops / ms / 1e9);
    CHECK(hipFree(out)); CHECK(hipFree(tab));
    return 0;
}

int main()
{
    for (int w : {1, 2, 3}) {
        run<0>("16 MFMA in place", w);
        run<1>("+ 4 ds_read_b128", w);
        run<2>("ping-pong accumulators", w);
        run<3>("+ ds_read + global prefetch + copy", w);
    }
    return 0;
}
