// Can the CPU store directly into device memory (large BAR)?  Tries fine-grained and ordinary device allocations in a
// child process each (a store into an inaccessible mapping is a SIGSEGV), then times the mailbox round trip with the
// mailbox in device memory.   hipcc --offload-arch=gfx950 -O3 -o bar_write bar_write.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>
#include <algorithm>

__global__ void k_sum(const uint32_t *p, uint32_t *out) { out[0] = p[0] + p[1]; }

__global__ void k_mailbox(volatile uint32_t *seq_dev, uint32_t *done_host, uint32_t last, long long idle_ticks)
{
    long long t_idle = wall_clock64();
    for (;;) {
        uint32_t s;
        for (;;) {
            s = __hip_atomic_load((uint32_t *)seq_dev, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (s != last) break;
            if (wall_clock64() - t_idle > idle_ticks) return;
            __builtin_amdgcn_s_sleep(2);
        }
        __hip_atomic_store(done_host, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        last = s;
        t_idle = wall_clock64();
    }
}

static int try_kind(int kind)
{
    uint32_t *d = nullptr, *out = nullptr;
    hipError_t e = kind == 0 ? hipExtMallocWithFlags((void **)&d, 4096, hipDeviceMallocFinegrained)
                 : kind == 1 ? hipMalloc((void **)&d, 4096)
                             : hipExtMallocWithFlags((void **)&d, 4096, hipDeviceMallocUncached);
    if (e != hipSuccess) { printf("kind %d: allocation failed: %s\n", kind, hipGetErrorString(e)); return 3; }
    if (hipHostMalloc((void **)&out, 64, hipHostMallocMapped) != hipSuccess) return 3;
    hipMemset(d, 0, 4096);
    hipDeviceSynchronize();
    volatile uint32_t *h = d;
    h[0] = 40; h[1] = 2;             // <- SIGSEGV here if the CPU cannot reach it
    __sync_synchronize();
    const uint32_t back = h[0];
    k_sum<<<1, 1>>>(d, out);
    hipDeviceSynchronize();
    printf("kind %d: CPU store ok, CPU load back %u, kernel saw sum %u\n", kind, back, out[0]);
    if (out[0] != 42) return 2;
    // mailbox round trip, mailbox word in device memory
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    h[16] = 0; out[0] = 0;
    k_mailbox<<<1, 1, 0, st>>>(d + 16, out, 0, 100000LL * 2);
    const int iters = 20000;
    std::vector<double> us(iters);
    volatile uint32_t *done = out;
    for (int i = 0; i < iters; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        h[16] = i + 1;
        __sync_synchronize();
        while (*done != (uint32_t)(i + 1)) { }
        us[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    std::sort(us.begin(), us.end());
    printf("kind %d: mailbox in device memory: round trip median %.2f us  p10 %.2f  p90 %.2f\n", kind, us[iters / 2], us[iters / 10], us[iters * 9 / 10]);
    hipStreamSynchronize(st);
    return 0;
}

int main()
{
    for (int kind = 0; kind < 3; ++kind) {
        fflush(stdout);
        const pid_t pid = fork();
        if (pid == 0) { alarm(20); const int rc = try_kind(kind); fflush(stdout); _exit(rc); }
        int st = 0;
        waitpid(pid, &st, 0);
        if (WIFSIGNALED(st)) printf("kind %d: child died with signal %d (CPU cannot reach this memory)\n", kind, WTERMSIG(st));
        else printf("kind %d: exit %d\n", kind, WEXITSTATUS(st));
    }
    return 0;
}
