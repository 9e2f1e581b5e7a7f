// Micro-benchmark: issue rate of the candidate inner-loop instructions on gfx950.
//   plain   v_fmac_f32 (VGPR operands)
//   dpp     v_fmac_f32_dpp row_newbcast
//   pk      v_pk_fma_f32 (VGPR operands)
//   pk_s    v_pk_fma_f32 with an SGPR-pair operand
//   fma_s   v_fmac_f32 with an SGPR operand
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, const float *in, int iters)
{
    float acc[16];
    float c = in[threadIdx.x & 63], x = in[64 + (threadIdx.x & 63)];
    float sc = __builtin_amdgcn_readfirstlane(in[1]), sc2 = __builtin_amdgcn_readfirstlane(in[2]);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = (float)r;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc[r]) : "v"(c), "v"(x));
            } else if (MODE == 1) {
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(acc[0]) : "v"(c), "v"(x));
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(acc[1]) : "v"(c), "v"(x));
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "+v"(acc[2]) : "v"(c), "v"(x));
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[3]) : "v"(c), "v"(x));
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf" : "+v"(acc[4]) : "v"(c), "v"(x));
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc[5]) : "v"(c), "v"(x));
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:6 row_mask:0xf bank_mask:0xf" : "+v"(acc[6]) : "v"(c), "v"(x));
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "+v"(acc[7]) : "v"(c), "v"(x));
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xf" : "+v"(acc[8]) : "v"(c), "v"(x));
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:9 row_mask:0xf bank_mask:0xf" : "+v"(acc[9]) : "v"(c), "v"(x));
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:10 row_mask:0xf bank_mask:0xf" : "+v"(acc[10]) : "v"(c), "v"(x));
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:11 row_mask:0xf bank_mask:0xf" : "+v"(acc[11]) : "v"(c), "v"(x));
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:12 row_mask:0xf bank_mask:0xf" : "+v"(acc[12]) : "v"(c), "v"(x));
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:13 row_mask:0xf bank_mask:0xf" : "+v"(acc[13]) : "v"(c), "v"(x));
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:14 row_mask:0xf bank_mask:0xf" : "+v"(acc[14]) : "v"(c), "v"(x));
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf" : "+v"(acc[15]) : "v"(c), "v"(x));
            } else if (MODE == 2) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 cc = {c, c}, xx = {x, x};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f2 a = {acc[r], acc[r + 1]};
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(cc), "v"(xx));
                    acc[r] = a.x; acc[r + 1] = a.y;
                }
            } else if (MODE == 3) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 ss = {sc, sc2}, xx = {x, x};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f2 a = {acc[r], acc[r + 1]};
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a) : "s"(ss), "v"(xx));
                    acc[r] = a.x; acc[r + 1] = a.y;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc[r]) : "s"(sc), "v"(x));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> int run(const char *name, int waves_per_simd)
{
    const int iters = 4000;
    const int blocks = 256 * waves_per_simd; // 256 threads = 4 waves = 1 per SIMD
    float *out, *in;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHECK(hipMalloc(&in, 1024));
    std::vector<float> h(256, 1e-3f);
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
    const double fma_per_lane = (double)iters * 8 * 16;
    const double flops = fma_per_lane * 2 * 64 * 4 * blocks;
    // cycles per wave-level "16 FMAs" at an assumed 2.4 GHz: waves per SIMD share the SIMD
    printf("%-6s waves/SIMD %d : %8.3f ms  %7.1f TFLOP/s\n", name, waves_per_simd, ms, flops / ms / 1e9);
    CHECK(hipFree(out)); CHECK(hipFree(in));
    return 0;
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("plain", w); run<1>("dpp", w); run<2>("pk", w); run<3>("pk_s", w); run<4>("fma_s", w);
    }
    return 0;
}
