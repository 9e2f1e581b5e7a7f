// valu_issue — how many waves per SIMD does gfx950 need to keep the fp32 vector ALU busy, for independent and for
// dependent instruction streams?  (The frequency-domain kernel's waves spend ~18 % of their life issuing VALU; whether
// one such wave can use a whole SIMD decides how many must be in their butterfly phase at once.)
//   mode 0: 16 independent v_fmac_f32 chains per lane (ILP 16)
//   mode 1: 4 independent chains (ILP 4)
//   mode 2: 1 dependent chain (ILP 1)
//   mode 3: add/sub/mul/fma mix shaped like a radix-4 butterfly (ILP ~8)
// Grid: 256 CUs x W waves per SIMD x 4 SIMDs, one workgroup of 64*4*W threads per CU.
// Build: hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void k(float *out, const float *in, int iters)
{
    float a[16];
    const float c = in[threadIdx.x & 63], x = in[64 + (threadIdx.x & 63)];
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = (float)r * c;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[r]) : "v"(c), "v"(x));
            } else if (MODE == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[r & 3]) : "v"(c), "v"(x));
            } else if (MODE == 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[0]) : "v"(c), "v"(x));
            } else if (MODE == 4) { // 16 independent packed FMAs per lane (8 register pairs, two rounds): 2 lane-operations each
                typedef float v2f __attribute__((ext_vector_type(2)));
                v2f *p = reinterpret_cast<v2f *>(a);
                const v2f cc = {c, c}, xx = {x, x};
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[r & 7]) : "v"(cc), "v"(xx));
            } else if (MODE == 5) { // packed add / sub mix shaped like radix-4 butterflies on complex pairs (ILP 4-8)
                typedef float v2f __attribute__((ext_vector_type(2)));
                v2f *p = reinterpret_cast<v2f *>(a);
#pragma unroll
                for (int h = 0; h < 8; h += 4) {
                    v2f s0, s1, d0, d1;
                    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(s0) : "v"(p[h + 0]), "v"(p[h + 2]));
                    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d0) : "v"(p[h + 0]), "v"(p[h + 2]));
                    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(s1) : "v"(p[h + 1]), "v"(p[h + 3]));
                    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d1) : "v"(p[h + 1]), "v"(p[h + 3]));
                    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[h + 0]) : "v"(s0), "v"(s1));
                    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(p[h + 2]) : "v"(s0), "v"(s1));
                    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(p[h + 1]) : "v"(d0), "v"(d1));
                    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(p[h + 3]) : "v"(d0), "v"(d1));
                }
            } else {
                // two radix-4 butterflies on (a0..a7) as complex pairs: 16 add/sub per butterfly, written so that
                // the compiler cannot fold it (asm)
#pragma unroll
                for (int h = 0; h < 16; h += 8) {
                    float s0, s1, d0, d1, t0, t1, e0, e1;
                    asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(s0) : "v"(a[h + 0]), "v"(a[h + 4]));
                    asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(s1) : "v"(a[h + 1]), "v"(a[h + 5]));
                    asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(d0) : "v"(a[h + 0]), "v"(a[h + 4]));
                    asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(d1) : "v"(a[h + 1]), "v"(a[h + 5]));
                    asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(t0) : "v"(a[h + 2]), "v"(a[h + 6]));
                    asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(t1) : "v"(a[h + 3]), "v"(a[h + 7]));
                    asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(e0) : "v"(a[h + 2]), "v"(a[h + 6]));
                    asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(e1) : "v"(a[h + 3]), "v"(a[h + 7]));
                    asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(a[h + 0]) : "v"(s0), "v"(t0));
                    asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(a[h + 1]) : "v"(s1), "v"(t1));
                    asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(a[h + 4]) : "v"(s0), "v"(t0));
                    asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(a[h + 5]) : "v"(s1), "v"(t1));
                    asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(a[h + 2]) : "v"(d0), "v"(e1));
                    asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(a[h + 3]) : "v"(d1), "v"(e0));
                    asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(a[h + 6]) : "v"(d0), "v"(e1));
                    asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(a[h + 7]) : "v"(d1), "v"(e0));
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += a[r];
    if (s == 1.2345f) out[threadIdx.x] = s;
}

int main()
{
    float *in, *out;
    CHECK(hipMalloc(&in, 4096));
    CHECK(hipMalloc(&out, 4096));
    CHECK(hipMemset(in, 0, 4096));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 4000;
    const char *names[] = {"ILP16 fmac", "ILP4 fmac", "ILP1 fmac", "radix-4 add/sub mix", "ILP8 pk_fma", "radix-4 pk_add mix"};
    for (int mode = 0; mode < 6; ++mode)
        for (int W : {1, 2, 3, 4}) {
            const dim3 grid(256), block(64 * 4 * W); // one workgroup per CU: W waves on each of its 4 SIMDs
            if (block.x > 1024) { // two workgroups per CU instead
                continue;
            }
            auto launch = [&]() {
                switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, grid, block, 0, nullptr, out, in, iters); break;
                case 1: hipLaunchKernelGGL(k<1>, grid, block, 0, nullptr, out, in, iters); break;
                case 2: hipLaunchKernelGGL(k<2>, grid, block, 0, nullptr, out, in, iters); break;
                case 3: hipLaunchKernelGGL(k<3>, grid, block, 0, nullptr, out, in, iters); break;
                case 4: hipLaunchKernelGGL(k<4>, grid, block, 0, nullptr, out, in, iters); break;
                default: hipLaunchKernelGGL(k<5>, grid, block, 0, nullptr, out, in, iters); break;
                }
            };
            launch();
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, nullptr));
            launch();
            CHECK(hipEventRecord(e1, nullptr));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double insts_per_wave = (double)iters * 4 * (mode == 3 ? 32 : 16); // (modes 4, 5: packed instructions, two lane-operations each)
            const double cyc_per_inst_per_simd = ms * 1e-3 * 2.4e9 / (insts_per_wave * W); // at a nominal 2.4 GHz
            printf("%-20s W=%d waves/SIMD: %7.1f us  %.2f nominal cycles per wave-instruction per SIMD (2.0 = peak)\n", names[mode], W,
                   ms * 1e3, cyc_per_inst_per_simd);
        }
    return 0;
}
