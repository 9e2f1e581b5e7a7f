// c2_probe — the MEMORY side of configs[2] (44.1k -> 16k, 8 channels interleaved, 60 s) on its own, in the access shapes a
// kernel could use.  750 blocks of 4410 frames (hop 3528) in, 1280 frames out per block; frames are 32 bytes.
//   mode 0: today's k_fft_strided2<CP>: workgroup = (block, channel pair), 320 threads, 35 KB LDS; thread j loads the 8-byte
//           word of frames j + 210 t (t < 21) — 8 bytes at a 32-byte stride — and stores 8-byte words at a 32-byte stride
//   mode 1: workgroup = (block, TWO channel pairs), 16-byte words at a 32-byte stride, 70 KB LDS, 320 threads
//   mode 2: workgroup = (block, all 8 channels), lanes run over (frame, half frame): 16-byte loads, fully contiguous;
//           141 KB LDS, 1024 threads; stores whole frames
//   mode 3: mode 0's loads, but lanes run over (frame, channel pair) inside a workgroup of 4 x 210 threads = (block, all
//           pairs): contiguous 8-byte loads; 141 KB LDS, 896 threads
// XCD-aware ids as in the product: the units of one block are adjacent in dispatch order on one XCD.
// Build: hipcc --offload-arch=gfx950 -O3 c2_probe.hip -o c2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int NA = 4410, HOP_IN = 3528, HOP_OUT = 1280, NBLK = 750, NB1 = 210;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float *__restrict__ out, const float *__restrict__ in, long long in_frames, long long out_frames)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int UNITS = MODE == 0 ? 4 : MODE == 1 ? 2 : 1;
    // XCD-aware: x = 8 * slot + xcd, slot = chunk * UNITS + unit, block = xcd * per_xcd + chunk
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, per_xcd = (NBLK + 7) / 8;
    const unsigned unit = slot % UNITS, chunk = slot / UNITS;
    if (chunk >= per_xcd) return;
    const unsigned blk = xcd * per_xcd + chunk;
    if (blk >= NBLK) return;
    const long long f0 = (long long)blk * HOP_IN, o0 = (long long)blk * HOP_OUT;
    float acc = 0.f;
    const int j = threadIdx.x;
    if (MODE == 0) {
        const v2f *x = reinterpret_cast<const v2f *>(in) + f0 * 4 + unit;
        if (j < NB1) {
            v2f v[21];
#pragma unroll
            for (int t = 0; t < 21; ++t) { const long long f = j + NB1 * t; v[t] = (f0 + f < in_frames) ? x[f * 4] : v2f{0.f, 0.f}; }
#pragma unroll
            for (int t = 0; t < 21; ++t) acc += v[t].x + v[t].y;
        }
    } else if (MODE == 1) {
        const v4f *x = reinterpret_cast<const v4f *>(in) + f0 * 2 + unit;
        if (j < NB1) {
            v4f v[21];
#pragma unroll
            for (int t = 0; t < 21; ++t) { const long long f = j + NB1 * t; v[t] = (f0 + f < in_frames) ? x[f * 2] : v4f{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int t = 0; t < 21; ++t) acc += v[t].x + v[t].y + v[t].z + v[t].w;
        }
    } else if (MODE == 2) {
        const v4f *x = reinterpret_cast<const v4f *>(in) + f0 * 2;
#pragma unroll
        for (int t = 0; t < 9; ++t) { // 8820 half frames over 1024 threads
            const int q = j + 1024 * t;
            if (q < NA * 2 && f0 + q / 2 < in_frames) { const v4f v = x[q]; acc += v.x + v.y + v.z + v.w; }
        }
    } else {
        const v2f *x = reinterpret_cast<const v2f *>(in) + f0 * 4;
        if (j < 4 * NB1) { // lane = 4 * butterfly + pair: a wave reads 16 whole frames per load
            v2f v[21];
#pragma unroll
            for (int t = 0; t < 21; ++t) { const long long f = (j >> 2) + NB1 * t; v[t] = (f0 + f < in_frames) ? x[f * 4 + (j & 3)] : v2f{0.f, 0.f}; }
#pragma unroll
            for (int t = 0; t < 21; ++t) acc += v[t].x + v[t].y;
        }
    }
    if (acc == 1.2345678f) reinterpret_cast<float *>(smem)[j] = acc;
    // stores
    if (MODE == 0) {
        v2f *y = reinterpret_cast<v2f *>(out) + o0 * 4 + unit;
        for (int f = j; f < HOP_OUT; f += blockDim.x) if (o0 + f < out_frames) y[(long long)f * 4] = v2f{acc, acc};
    } else if (MODE == 1) {
        v4f *y = reinterpret_cast<v4f *>(out) + o0 * 2 + unit;
        for (int f = j; f < HOP_OUT; f += blockDim.x) if (o0 + f < out_frames) y[(long long)f * 2] = v4f{acc, acc, acc, acc};
    } else {
        v4f *y = reinterpret_cast<v4f *>(out) + o0 * 2;
        for (int q = j; q < HOP_OUT * 2; q += blockDim.x) if (o0 + q / 2 < out_frames) __builtin_nontemporal_store(v4f{acc, acc, acc, acc}, y + q);
    }
}

int main()
{
    const long long in_frames = 2646000, out_frames = 960000;
    const int SETS = 3;
    float *in[SETS], *out[SETS];
    for (int s = 0; s < SETS; ++s) {
        CHECK(hipMalloc((void **)&in[s], in_frames * 32 + 4096)); CHECK(hipMalloc((void **)&out[s], out_frames * 32 + 4096));
        CHECK(hipMemset(in[s], 0x3c, in_frames * 32)); CHECK(hipMemset(out[s], 0, out_frames * 32));
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct { int mode, units, nt; size_t lds; const char *name; } cfg[] = {
        {0, 4, 320, 35280, "8 B @ 32 B stride, (block, pair), 35 KB"}, {1, 2, 320, 70560, "16 B @ 32 B stride, (block, 2 pairs), 70 KB"},
        {2, 1, 1024, 141120, "16 B contiguous, (block, 8 ch), 141 KB, 1024 thr"}, {3, 1, 896, 141120, "8 B contiguous (lane = 4 bfly + pair), 141 KB, 896 thr"},
        {0, 4, 320, 0, "mode 0 without the LDS footprint"}, {2, 1, 1024, 0, "mode 2 without the LDS footprint"}};
    for (auto &c : cfg) {
        const unsigned grid = (unsigned)(((NBLK + 7) / 8) * 8 * c.units);
        auto launch = [&](int s) {
            switch (c.mode) {
            case 0: hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); hipLaunchKernelGGL(k<0>, dim3(grid), dim3(c.nt), c.lds, 0, out[s], in[s], in_frames, out_frames); break;
            case 1: hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); hipLaunchKernelGGL(k<1>, dim3(grid), dim3(c.nt), c.lds, 0, out[s], in[s], in_frames, out_frames); break;
            case 2: hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); hipLaunchKernelGGL(k<2>, dim3(grid), dim3(c.nt), c.lds, 0, out[s], in[s], in_frames, out_frames); break;
            default: hipFuncSetAttribute((const void *)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); hipLaunchKernelGGL(k<3>, dim3(grid), dim3(c.nt), c.lds, 0, out[s], in[s], in_frames, out_frames); break;
            }
        };
        for (int i = 0; i < 5; ++i) launch(i % SETS);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, nullptr));
        const int reps = 60;
        for (int i = 0; i < reps; ++i) launch(i % SETS);
        CHECK(hipEventRecord(e1, nullptr));
        CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("mode %d  %-58s %7.2f us per launch  (%.2f TB/s of the 115.4 MB)\n", c.mode, c.name, ms * 1e3 / reps, 115.392e6 / (ms * 1e-3 / reps) / 1e12);
    }
    return 0;
}
