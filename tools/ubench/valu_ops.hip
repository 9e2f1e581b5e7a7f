// valu_ops — issue cost of the fp32 VALU forms the frequency-domain kernel is made of, by number of DISTINCT vector
// source operands: does a 3-source v_fma_f32 issue at the 2-source rate?  (valu_issue.hip showed 16 independent
// v_fmac with two shared sources at ~4 cycles per wave-instruction where a dependent chain reaches 2.3.)
//   mode 0: v_add_f32   d = a + b                (2 distinct VGPR sources)
//   mode 1: v_mul_f32   d = a * b
//   mode 2: v_fma_f32   d = a * b + c            (3 distinct VGPR sources)
//   mode 3: v_fmac_f32  d += a * b               (d, a, b distinct, d changes per instruction)
//   mode 4: v_fma_f32   d = a * s + c            (one source an SGPR)
//   mode 5: complex multiply as the compiler emits it: mul, mul, fma, fma
//   mode 6: v_fma_f32   d = a * a + c            (2 distinct VGPRs, 3 operands)
//   mode 7: v_pk_add_f32
//   mode 8: v_fma_f32 d = a*b + c with c = the previous instruction's result of another chain (forwarding?)
// W waves per SIMD (one workgroup of 256 W threads per CU), 8 independent destination registers per lane.
// Build: hipcc --offload-arch=gfx950 -O3 valu_ops.hip -o valu_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float *out, const float *in, int iters)
{
    float d[8], a[8], b[8], c[8];
    const float s = in[200];
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(in[201 + (threadIdx.x & 63)] > 0.f);
#pragma unroll
    for (int r = 0; r < 8; ++r) { d[r] = in[(threadIdx.x + r) & 63]; a[r] = in[64 + ((threadIdx.x + r) & 63)]; b[r] = in[128 + ((threadIdx.x + 3 * r) & 63)]; c[r] = in[(threadIdx.x + 5 * r) & 63]; }
    v2f pd[4], pa[4], pb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { pd[r] = v2f{d[2 * r], d[2 * r + 1]}; pa[r] = v2f{a[2 * r], a[2 * r + 1]}; pb[r] = v2f{b[2 * r], b[2 * r + 1]}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (MODE == 0) asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(d[r]) : "v"(a[r]), "v"(b[r]));
                if (MODE == 1) asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(d[r]) : "v"(a[r]), "v"(b[r]));
                if (MODE == 2) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d[r]) : "v"(a[r]), "v"(b[r]), "v"(c[r]));
                if (MODE == 3) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(d[r]) : "v"(a[r]), "v"(b[r]));
                if (MODE == 4) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d[r]) : "v"(a[r]), "s"(s), "v"(c[r]));
                if (MODE == 6) asm volatile("v_fma_f32 %0, %1, %1, %2" : "=v"(d[r]) : "v"(a[r]), "v"(c[r]));
                if (MODE == 9) asm volatile("v_fmac_f32_e32 %0, 0x3f6c835e, %1" : "+v"(d[r]) : "v"(a[r]));
                if (MODE == 10) asm volatile("v_mul_f32_e32 %0, 0x3f6c835e, %1" : "=v"(d[r]) : "v"(a[r]));
                if (MODE == 11) asm volatile("v_fmamk_f32 %0, %1, 0x3f6c835e, %2" : "=v"(d[r]) : "v"(a[r]), "v"(b[r]));
                if (MODE == 12) asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(d[r]) : "s"(s), "v"(a[r]));
                if (MODE == 13) asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(d[r]) : "s"(s), "v"(a[r]));
                if (MODE == 14) asm volatile("v_mul_f32_e32 %0, 0.5, %1" : "=v"(d[r]) : "v"(a[r]));
                if (MODE == 15) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(d[r]) : "s"(s), "v"(a[r]));
                if (MODE == 16) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(d[r]) : "v"(c[0]), "v"(a[r]));
                if (MODE == 17) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d[r]) : "v"(a[r]), "v"(b[r]), "s"(mask));
                if (MODE == 21) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(d[r]) : "v"(c[0]), "v"(a[r]), "v"(b[r]));
                if (MODE == 8) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d[r]) : "v"(a[r]), "v"(b[r]), "v"(d[(r + 7) & 7]));
            }
            if (MODE == 5) {
#pragma unroll
                for (int r = 0; r < 8; r += 2) { // (d[r], d[r+1]) = (a[r], a[r+1]) * (b[r], b[r+1])
                    float t0, t1;
                    asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(t0) : "v"(a[r + 1]), "v"(b[r + 1]));
                    asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(t1) : "v"(a[r + 1]), "v"(b[r]));
                    asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(d[r]) : "v"(a[r]), "v"(b[r]), "v"(t0));
                    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d[r + 1]) : "v"(a[r]), "v"(b[r + 1]), "v"(t1));
                }
#pragma unroll
                for (int r = 0; r < 8; r += 2) {
                    float t0, t1;
                    asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(t0) : "v"(c[r + 1]), "v"(b[r + 1]));
                    asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(t1) : "v"(c[r + 1]), "v"(b[r]));
                    asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(a[r]) : "v"(c[r]), "v"(b[r]), "v"(t0));
                    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[r + 1]) : "v"(c[r]), "v"(b[r + 1]), "v"(t1));
                }
            }
            if (MODE == 18) // (one statement: the compiler must not touch vcc between the move and its readers)
                asm volatile("s_mov_b64 vcc, %8\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_cndmask_b32_e32 %1, %1, %2, vcc\n\tv_cndmask_b32_e32 %2, %2, %3, vcc\n\t"
                             "v_cndmask_b32_e32 %3, %3, %4, vcc\n\tv_cndmask_b32_e32 %4, %4, %5, vcc\n\tv_cndmask_b32_e32 %5, %5, %6, vcc\n\t"
                             "v_cndmask_b32_e32 %6, %6, %7, vcc\n\tv_cndmask_b32_e32 %7, %7, %0, vcc"
                             : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "s"(mask) : "vcc");
            if (MODE == 19 || MODE == 20) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (MODE == 19) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(pd[r]) : "v"(pa[r]), "v"(pb[r]));
                    if (MODE == 20) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pd[r]) : "v"(pa[r]), "v"(pb[r]));
                }
            }
            if (MODE == 7) {
#pragma unroll
                for (int r = 0; r < 4; ++r) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(pd[r]) : "v"(pa[r]), "v"(pb[r]));
#pragma unroll
                for (int r = 0; r < 4; ++r) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(pa[r]) : "v"(pd[r]), "v"(pb[r]));
            }
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) sum += d[r] + a[r];
#pragma unroll
    for (int r = 0; r < 4; ++r) sum += pd[r].x + pd[r].y + pa[r].x;
    if (sum == 1.2345f) out[threadIdx.x] = sum;
}

template <int MODE> int run(float *out, const float *in, const char *name, int per_iter)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int W : {1, 2, 4}) {
        const dim3 grid(256), block(64 * 4 * W);
        hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, out, in, 10);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, out, in, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double instr = (double)iters * per_iter * W; // wave-instructions per SIMD
        printf("%-44s W=%d: %8.1f us  %.2f nominal cycles (2.4 GHz) per wave-instruction per SIMD\n", name, W, ms * 1e3, ms * 1e-3 * 2.4e9 / instr);
    }
    return 0;
}

int main()
{
    float *in, *out;
    CHECK(hipMalloc(&in, 4096)); CHECK(hipMalloc(&out, 4096)); CHECK(hipMemset(in, 0, 4096));
    run<0>(out, in, "v_add_f32 (2 VGPR sources)", 64);
    run<1>(out, in, "v_mul_f32 (2 VGPR sources)", 64);
    run<2>(out, in, "v_fma_f32 (3 distinct VGPR sources)", 64);
    run<3>(out, in, "v_fmac_f32 (dst + 2 VGPR sources)", 64);
    run<4>(out, in, "v_fma_f32 (2 VGPR + 1 SGPR source)", 64);
    run<6>(out, in, "v_fma_f32 a*a+c (2 distinct VGPRs)", 64);
    run<8>(out, in, "v_fma_f32 a*b+prev (3 VGPRs, one just written)", 64);
    run<9>(out, in, "v_fmac_f32 d += LITERAL * a", 64);
    run<10>(out, in, "v_mul_f32 d = LITERAL * a", 64);
    run<11>(out, in, "v_fmamk_f32 d = a * LITERAL + b", 64);
    run<12>(out, in, "v_mul_f32 d = SGPR * a", 64);
    run<13>(out, in, "v_add_f32 d = SGPR + a", 64);
    run<14>(out, in, "v_mul_f32 d = 0.5 (inline) * a", 64);
    run<15>(out, in, "v_fmac_f32 d += SGPR * a", 64);
    run<16>(out, in, "v_fmac_f32 d += c0 (shared VGPR) * a", 64);
    run<5>(out, in, "complex multiply (mul mul fma fma)", 8 * 32 / 8 * 8 / 8 * 8); // 32 per u, 8 u's
    run<7>(out, in, "v_pk_add_f32", 64);
    run<17>(out, in, "v_cndmask_b32_e64 (mask in an SGPR pair)", 64);
    run<18>(out, in, "v_cndmask_b32_e32 (mask in vcc; + 1 s_mov per 8)", 64);
    run<21>(out, in, "v_bfi_b32 (mask in a VGPR)", 64);
    run<19>(out, in, "v_pk_fma_f32 op_sel broadcast (per pk instr)", 64);
    run<20>(out, in, "v_pk_fma_f32 plain (per pk instr)", 64);
    return 0;
}
