// permlane_swap — what v_permlane32_swap_b32 does on gfx950, and whether 8-byte buffer loads work at 4-byte alignment
// (csrc/fftwave.hip relies on both).  Build: hipcc --offload-arch=gfx950 -O3 permlane_swap.hip -o permlane_swap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *o, const float *x, float *y)
{
    unsigned a = threadIdx.x, b = threadIdx.x + 100;
    v2u r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[threadIdx.x] = r.x; o[64 + threadIdx.x] = r.y;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(x + 1), 0, 4096, 0x00020000); // base 4-byte aligned only
    v2u p = __builtin_amdgcn_raw_buffer_load_b64(rs, threadIdx.x * 8, 0, 0);
    { const unsigned a0 = p.x, a1 = p.y; y[2 * threadIdx.x] = __builtin_bit_cast(float, a0); y[2 * threadIdx.x + 1] = __builtin_bit_cast(float, a1); } // (by value: bit_cast of `p.y` itself reads element 0)
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void *)(x + 2), 0, 4096, 0x00020000); // 8-byte aligned base
    v2u q = __builtin_amdgcn_raw_buffer_load_b64(r2, threadIdx.x * 8, 0, 0);
    { const unsigned a0 = q.x, a1 = q.y; y[128 + 2 * threadIdx.x] = __builtin_bit_cast(float, a0); y[128 + 2 * threadIdx.x + 1] = __builtin_bit_cast(float, a1); }
    const float2 g = *reinterpret_cast<const float2 *>(x + 1 + 2 * threadIdx.x); // plain 8-byte global load at 4-byte alignment
    y[256 + 2 * threadIdx.x] = g.x; y[256 + 2 * threadIdx.x + 1] = g.y;
}
// the load pattern of csrc/fftwave.hip: lane l loads x[2 l + 128 t'], x[2 l + 1 + 128 t'] and one swap per pair leaves it with
// column ((l & 31) << 1 | l >> 5) at t = 2 t', 2 t' + 1
__global__ void k2(const float *x, float *out)
{
    const int lane = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, 8192, 0x00020000);
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
        v2u p = __builtin_amdgcn_raw_buffer_load_b64(rs, lane * 8, 128 * tp * 4, 0);
        v2u s = __builtin_amdgcn_permlane32_swap(p.x, p.y, false, false);
        const unsigned s0 = s.x, s1 = s.y;
        out[lane * 8 + 2 * tp] = __builtin_bit_cast(float, s0);
        out[lane * 8 + 2 * tp + 1] = __builtin_bit_cast(float, s1);
    }
}
int main()
{
    unsigned *o; float *x, *y;
    hipMalloc(&o, 512); hipMalloc(&x, 8192); hipMalloc(&y, 2048);
    float hx[2048]; for (int i = 0; i < 2048; ++i) hx[i] = (float)i;
    hipMemcpy(x, hx, 8192, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, x, y);
    unsigned ho[128]; float hy[384];
    hipMemcpy(ho, o, 512, hipMemcpyDeviceToHost); hipMemcpy(hy, y, 1536, hipMemcpyDeviceToHost);
    printf("first result  (old = lane):       lane 0 %u lane 31 %u lane 32 %u lane 63 %u\n", ho[0], ho[31], ho[32], ho[63]);
    printf("second result (old = lane + 100): lane 0 %u lane 31 %u lane 32 %u lane 63 %u\n", ho[64], ho[95], ho[96], ho[127]);
    bool ok = true; for (int i = 0; i < 128; ++i) ok = ok && hy[i] == (float)(i + 1);
    printf("8-byte buffer loads from a base that is only 4-byte aligned: %s (y[0..3] = %g %g %g %g)\n", ok ? "correct" : "WRONG", hy[0], hy[1], hy[2], hy[3]);
    ok = true; for (int i = 0; i < 128; ++i) ok = ok && hy[128 + i] == (float)(i + 2);
    printf("8-byte buffer loads from an 8-byte aligned base: %s (%g %g %g %g)\n", ok ? "correct" : "WRONG", hy[128], hy[129], hy[130], hy[131]);
    ok = true; for (int i = 0; i < 128; ++i) ok = ok && hy[256 + i] == (float)(i + 1);
    printf("8-byte global loads at 4-byte alignment: %s (%g %g %g %g)\n", ok ? "correct" : "WRONG", hy[256], hy[257], hy[258], hy[259]);
    float *o2; hipMalloc(&o2, 2048);
    hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, x, o2);
    float h2[512]; hipMemcpy(h2, o2, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int t = 0; t < 8; ++t) bad += h2[l * 8 + t] != (float)((((l & 31) << 1) | (l >> 5)) + 64 * t);
    printf("pair loads + swap = column form: %d wrong of 512 (lane 0: %g %g %g %g; lane 33: %g %g %g %g)\n", bad, h2[0], h2[1], h2[2], h2[3], h2[33 * 8], h2[33 * 8 + 1], h2[33 * 8 + 2], h2[33 * 8 + 3]);
    return 0;
}
