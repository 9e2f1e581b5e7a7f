// Round-trip latency of a host <-> resident-kernel mailbox in pinned host memory (no HIP call per message):
// host writes seq, kernel (one workgroup, lane 0 polling) answers by writing done = seq; the kernel leaves
// after `idle_us` without a message (and after `life_us` whatever happens).
// Also: the same with a payload read from pinned memory (n bytes) before answering.
// hipcc --offload-arch=gfx950 -O3 -o mailbox mailbox.hip
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

struct Box { uint32_t seq; uint32_t n_bytes; uint32_t pad[14]; uint32_t done; uint32_t exited; uint32_t pad2[14]; };

__global__ void k_mailbox(Box *box, const uint32_t *payload, uint32_t *sink_host, uint32_t last, long long idle_ticks, long long life_ticks)
{
    __shared__ uint32_t s_seq, s_n;
    const long long t_start = wall_clock64();
    long long t_idle = t_start;
    for (;;) {
        if (threadIdx.x == 0) {
            uint32_t s;
            for (;;) {
                s = __hip_atomic_load(&box->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                const long long now = wall_clock64();
                if (s != last) break;
                if (now - t_idle > idle_ticks || now - t_start > life_ticks) { s = last; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            s_seq = s;
            s_n = __hip_atomic_load(&box->n_bytes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
        const uint32_t s = s_seq, n = s_n;
        if (s == last) break;
        uint32_t acc = 0;
        for (uint32_t i = threadIdx.x; i < n / 4; i += blockDim.x)
            acc += __hip_atomic_load(&payload[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (n) sink_host[threadIdx.x] = acc + s; // "result" into pinned memory
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&box->done, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        last = s;
        t_idle = wall_clock64();
        __syncthreads();
    }
    if (threadIdx.x == 0) __hip_atomic_store(&box->exited, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    Box *box; uint32_t *payload, *sink;
    CK(hipHostMalloc(&box, sizeof(Box), hipHostMallocMapped));
    CK(hipHostMalloc(&payload, 1 << 16, hipHostMallocMapped));
    CK(hipHostMalloc(&sink, 4096, hipHostMallocMapped));
    memset(box, 0, sizeof *box); memset(payload, 1, 1 << 16);
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int clk_khz = 100000; // wall_clock64: 100 MHz on gfx9
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0);
    printf("wall clock %d kHz\n", clk_khz);
    const long long idle = (long long)clk_khz * 2;      // 2 ms
    const long long life = (long long)clk_khz * 3000;   // 3 s
    for (uint32_t nbytes : {0u, 1024u, 4096u}) {
        box->exited = 0; box->n_bytes = nbytes;
        const uint32_t base = box->seq;
        hipLaunchKernelGGL(k_mailbox, dim3(1), dim3(256), 0, st, box, payload, sink, base, idle, life);
        CK(hipGetLastError());
        std::vector<double> us(iters);
        volatile uint32_t *done = &box->done;
        for (int i = 0; i < iters; ++i) {
            const auto t0 = std::chrono::steady_clock::now();
            __atomic_store_n(&box->seq, base + 1 + i, __ATOMIC_RELEASE);
            while (*done != base + 1 + i) {
                if (__atomic_load_n(&box->exited, __ATOMIC_ACQUIRE)) { printf("kernel left early at %d\n", i); return 2; }
            }
            us[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        }
        std::sort(us.begin(), us.end());
        printf("payload %5u B: round trip median %.2f us  p10 %.2f  p90 %.2f  p99 %.2f\n", nbytes, us[iters / 2], us[iters / 10], us[iters * 9 / 10], us[iters * 99 / 100]);
        const auto t0 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(st)); // the kernel leaves by itself after 2 ms of silence
        printf("   kernel left %.2f ms after the last message, exited flag %u\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), box->exited);
    }
    return 0;
}
