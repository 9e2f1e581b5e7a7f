// valu_power — board power of the fp32 vector ALU at equal flop rate: scalar v_add/v_fma against packed v_pk_add/v_pk_fma
// (MI355X batch launches run AT the 1.4 kW cap, so time = energy / cap: a form that moves the same flops with fewer
// instructions is worth exactly its power saving).  Runs each mode ~3 s on random data; sample rocm-smi beside it:
//   tools/ubench/valu_power <mode>     (tools/valu_power.sh drives it)
//   mode 0: 2 x v_add_f32    mode 1: 1 x v_pk_add_f32    mode 2: 2 x v_fma_f32 (3 sources)    mode 3: 1 x v_pk_fma_f32
//   mode 4: cmul as mul mul fma fma    mode 5: cmul as v_pk_mul + v_pk_fma (op_sel forms)    mode 6: idle spin (s_sleep)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(512) k(float *out, const float *in, int iters)
{
    v2f d[8], a[8], b[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        d[r] = v2f{in[(threadIdx.x * 7 + r) & 1023], in[(threadIdx.x * 3 + r + 1) & 1023]};
        a[r] = v2f{in[(threadIdx.x * 5 + 2 * r) & 1023], in[(threadIdx.x + 9 * r) & 1023]};
        b[r] = v2f{in[(threadIdx.x * 11 + r) & 1023], in[(threadIdx.x * 13 + 3 * r) & 1023]};
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (MODE == 0) {
                    asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(d[r].x) : "v"(a[r].x), "v"(b[r].x));
                    asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(d[r].y) : "v"(a[r].y), "v"(b[r].y));
                }
                if (MODE == 1) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d[r]) : "v"(a[r]), "v"(b[r]));
                if (MODE == 2) {
                    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d[r].x) : "v"(a[r].x), "v"(b[r].x), "v"(a[(r + 1) & 7].y));
                    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d[r].y) : "v"(a[r].y), "v"(b[r].y), "v"(a[(r + 1) & 7].x));
                }
                if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d[r]) : "v"(a[r]), "v"(b[r]), "v"(a[(r + 1) & 7]));
                if (MODE == 4) {
                    float t0, t1;
                    asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(t0) : "v"(a[r].y), "v"(b[r].y));
                    asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(t1) : "v"(a[r].y), "v"(b[r].x));
                    asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(d[r].x) : "v"(a[r].x), "v"(b[r].x), "v"(t0));
                    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d[r].y) : "v"(a[r].x), "v"(b[r].y), "v"(t1));
                }
                if (MODE == 5) {
                    v2f t;
                    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a[r]), "v"(b[r]));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(d[r]) : "v"(a[r]), "v"(b[r]), "v"(t));
                }
                if (MODE == 6) __builtin_amdgcn_s_sleep(8);
            }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) a[r] = a[r] * 0.999f + d[r] * 0.001f; // (keeps values bounded and data-dependent)
    }
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) sum += d[r].x + d[r].y + a[r].x;
    if (sum == 1.2345f) out[threadIdx.x] = sum;
}

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    float *in, *out, h[1024];
    srand(1);
    for (int i = 0; i < 1024; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    CHECK(hipMalloc(&in, 4096)); CHECK(hipMalloc(&out, 4096)); CHECK(hipMemcpy(in, h, 4096, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const dim3 grid(256 * 4), block(512); // 4 workgroups of 8 waves per CU: 8 waves per SIMD
    const int iters = 20000;
    void (*kern)(float *, const float *, int) = mode == 0 ? k<0> : mode == 1 ? k<1> : mode == 2 ? k<2> : mode == 3 ? k<3> : mode == 4 ? k<4> : mode == 5 ? k<5> : k<6>;
    CHECK(hipEventRecord(e0));
    int n = 0;
    for (; n < 60; ++n) hipLaunchKernelGGL(kern, grid, block, 0, 0, out, in, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double units = (double)n * iters * 32 * 8; // per SIMD: (pairs of scalar ops | packed ops) x waves
    printf("mode %d: %.0f ms total, %.2f nominal cycles (2.4 GHz) per complex-lane op pair per SIMD\n", mode, ms, ms * 1e-3 * 2.4e9 / units);
    return 0;
}
