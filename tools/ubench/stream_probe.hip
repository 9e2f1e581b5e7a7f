// stream_probe — what this box's HBM delivers to plain streaming kernels (the ceiling bench.py quotes
// roofline fractions against besides the 8 TB/s spec).  Not part of the product library.
//
// Variants (all 16 bytes per lane per access, 256-thread workgroups):
//   copy1    one float4 per thread, grid = n/4/256 workgroups (no loop)
//   copyU    U float4 per thread, all loads issued before the first store (U = 2, 4, 8), workgroup-contiguous tiles
//   copyUnt  the same with non-temporal loads and stores
//   read     U loads per thread, xor-reduced, one store per workgroup (read-only sweep)
//   write    U stores per thread (write-only sweep)
// Build: hipcc --offload-arch=gfx950 -O3 stream_probe.hip -o stream_probe            (executable: prints a table)
//        hipcc --offload-arch=gfx950 -O3 -shared -fPIC -DPROBE_LIB stream_probe.hip -o libstream_probe.so
//        (bench.py loads the library through ctypes: stream_probe_run(mode, dst, src, bytes, stream))
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ void __launch_bounds__(256) k_copy(v4f *__restrict__ dst, const v4f *__restrict__ src, size_t n16)
{
    // workgroup b owns the contiguous tile [b*256*U, (b+1)*256*U) of 16-byte words; access u of lane t is word
    // tile + u*256 + t: every wave-instruction moves one contiguous KiB
    const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < n16) v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < n16) {
            if (NT) __builtin_nontemporal_store(v[u], dst + i);
            else dst[i] = v[u];
        }
    }
}

template <int U>
__global__ void __launch_bounds__(256) k_read(float *__restrict__ sink, const v4f *__restrict__ src, size_t n16)
{
    const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * 256;
        v[u] = i < n16 ? src[i] : v4f{0.f, 0.f, 0.f, 0.f};
    }
    float a = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) a += v[u].x + v[u].y + v[u].z + v[u].w;
    if (a == 1.2345678f) sink[blockIdx.x] = a; // (never true for the probe's data: keeps the loads alive)
}

template <int U>
__global__ void __launch_bounds__(256) k_write(v4f *__restrict__ dst, size_t n16, float val)
{
    const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < n16) dst[i] = v4f{val, val, val, val};
    }
}

// mode: 0 copy1 | 1 copy2 | 2 copy4 | 3 copy8 | 4 copy4 nt | 5 copy8 nt | 6 read8 | 7 write4 | 8 read4
extern "C" __attribute__((visibility("default"))) int stream_probe_run(int mode, void *dst, const void *src, size_t bytes, void *stream)
{
    const size_t n16 = bytes / 16;
    hipStream_t st = (hipStream_t)stream;
    auto grid = [&](int U) { return dim3((unsigned)((n16 + 256 * (size_t)U - 1) / (256 * (size_t)U))); };
    switch (mode) {
    case 0: hipLaunchKernelGGL((k_copy<1, false>), grid(1), dim3(256), 0, st, (v4f *)dst, (const v4f *)src, n16); break;
    case 1: hipLaunchKernelGGL((k_copy<2, false>), grid(2), dim3(256), 0, st, (v4f *)dst, (const v4f *)src, n16); break;
    case 2: hipLaunchKernelGGL((k_copy<4, false>), grid(4), dim3(256), 0, st, (v4f *)dst, (const v4f *)src, n16); break;
    case 3: hipLaunchKernelGGL((k_copy<8, false>), grid(8), dim3(256), 0, st, (v4f *)dst, (const v4f *)src, n16); break;
    case 4: hipLaunchKernelGGL((k_copy<4, true>), grid(4), dim3(256), 0, st, (v4f *)dst, (const v4f *)src, n16); break;
    case 5: hipLaunchKernelGGL((k_copy<8, true>), grid(8), dim3(256), 0, st, (v4f *)dst, (const v4f *)src, n16); break;
    case 6: hipLaunchKernelGGL((k_read<8>), grid(8), dim3(256), 0, st, (float *)dst, (const v4f *)src, n16); break;
    case 7: hipLaunchKernelGGL((k_write<4>), grid(4), dim3(256), 0, st, (v4f *)dst, n16, 0.5f); break;
    case 8: hipLaunchKernelGGL((k_read<4>), grid(4), dim3(256), 0, st, (float *)dst, (const v4f *)src, n16); break;
    default: return -1;
    }
    return (int)hipGetLastError();
}

extern "C" __attribute__((visibility("default"))) const char *stream_probe_name(int mode)
{
    static const char *names[] = {"copy1", "copy2", "copy4", "copy8", "copy4_nt", "copy8_nt", "read8", "write4", "read4"};
    return mode >= 0 && mode < 9 ? names[mode] : nullptr;
}
// bytes moved through HBM per byte of `bytes`: 2 for copies, 1 for sweeps
extern "C" __attribute__((visibility("default"))) int stream_probe_moves(int mode) { return mode <= 5 ? 2 : 1; }

// ---- the MEMORY side of a block-transform launch on its own (bench.py `roofline.floor_us`): workgroup (bx, col) reads
// `wg_in` floats starting at col * in_col + bx * hop_in (4-byte loads, lane-contiguous: the paired FFT kernels' own load shape)
// and writes `wg_out` floats at col * out_col + bx * hop_out as 16-byte stores; `lds` bytes of dynamic LDS are declared so
// that as many workgroups fit a CU as in the real kernel.  No arithmetic.  (tools/ubench/pattern_probe.hip is the
// batch-only ancestor of this.)
struct PatternArgs { float *out; const float *in; long long in_col, out_col, in_len, out_len; int hop_in, hop_out, wg_in, wg_out; };
template <int NT>
__global__ void __launch_bounds__(NT) k_pattern(PatternArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const float *x = a.in + (long long)blockIdx.y * a.in_col + (long long)blockIdx.x * a.hop_in;
    const long long left_in = a.in_len - (long long)blockIdx.x * a.hop_in;
    float acc = 0.f;
    for (int i = threadIdx.x; i < a.wg_in; i += 8 * NT) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int n = i + u * NT; v[u] = (n < a.wg_in && n < left_in) ? x[n] : 0.f; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc == 1.2345678f) reinterpret_cast<float *>(smem)[threadIdx.x] = acc; // (never: keeps LDS and the loads alive)
    float *y = a.out + (long long)blockIdx.y * a.out_col + (long long)blockIdx.x * a.hop_out;
    const long long left_out = a.out_len - (long long)blockIdx.x * a.hop_out;
    const int head = (int)((4 - ((reinterpret_cast<uintptr_t>(y) >> 2) & 3)) & 3); // elements in front of the first whole granule
    for (int q = threadIdx.x; q * 4 + head + 3 < a.wg_out && q * 4 + head + 3 < left_out; q += NT)
        __builtin_nontemporal_store(v4f{acc, acc, acc, acc}, reinterpret_cast<v4f *>(y + head) + q);
}
extern "C" __attribute__((visibility("default"))) int stream_probe_pattern(void *dst, const void *src, unsigned grid_x, unsigned grid_y,
    int hop_in, int hop_out, int wg_in, int wg_out, long long in_col, long long out_col, long long in_len, long long out_len,
    unsigned lds, unsigned threads, void *stream)
{
    PatternArgs a{(float *)dst, (const float *)src, in_col, out_col, in_len, out_len, hop_in, hop_out, wg_in, wg_out};
    hipStream_t st = (hipStream_t)stream;
    if (threads == 384) hipLaunchKernelGGL((k_pattern<384>), dim3(grid_x, grid_y), dim3(384), lds, st, a);
    else if (threads == 320) hipLaunchKernelGGL((k_pattern<320>), dim3(grid_x, grid_y), dim3(320), lds, st, a);
    else hipLaunchKernelGGL((k_pattern<256>), dim3(grid_x, grid_y), dim3(256), lds, st, a);
    return (int)hipGetLastError();
}
// an empty launch: the floor under any one-kernel step (bench.py `launch_floor_us`)
__global__ void k_empty() {}
extern "C" __attribute__((visibility("default"))) int stream_probe_empty(void *stream)
{
    hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, (hipStream_t)stream);
    return (int)hipGetLastError();
}

#ifndef PROBE_LIB
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char **argv)
{
    const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : 1024) << 20; // MiB, default 1 GiB (well past the 256 MiB Infinity Cache)
    void *a, *b;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMemset(a, 0x3c, bytes));
    CHECK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("stream_probe: %zu MiB per buffer\n", bytes >> 20);
    for (int mode = 0; mode < 9; ++mode) {
        for (int i = 0; i < 3; ++i) stream_probe_run(mode, b, a, bytes, nullptr);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f, sum = 0.f;
        const int reps = 10;
        for (int r = 0; r < reps; ++r) {
            CHECK(hipEventRecord(e0, nullptr));
            stream_probe_run(mode, b, a, bytes, nullptr);
            CHECK(hipEventRecord(e1, nullptr));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
            sum += ms;
        }
        const double moved = (double)bytes * stream_probe_moves(mode);
        printf("  %-9s  avg %8.1f us  %7.1f GB/s   best %8.1f us  %7.1f GB/s\n", stream_probe_name(mode), sum / reps * 1e3,
               moved / (sum / reps * 1e-3) / 1e9, best * 1e3, moved / (best * 1e-3) / 1e9);
    }
    return 0;
}
#endif
