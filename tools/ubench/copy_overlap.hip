// Can a pageable H2D copy and a pageable D2H copy, issued from two host threads on two streams,
// overlap on this stack?  (Decides whether soxr.resample's host path is worth pipelining.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = 11520000;
    std::vector<char> hin(n, 1), hout(n, 0);
    void *din, *dout;
    hipMalloc(&din, n); hipMalloc(&dout, n);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        hipMemcpyAsync(din, hin.data(), n, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1);
        double t1 = now();
        hipMemcpyAsync(hout.data(), dout, n, hipMemcpyDeviceToHost, s2); hipStreamSynchronize(s2);
        double t2 = now();
        std::thread th([&] { hipMemcpyAsync(hout.data(), dout, n, hipMemcpyDeviceToHost, s2); hipStreamSynchronize(s2); });
        hipMemcpyAsync(din, hin.data(), n, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1);
        th.join();
        double t3 = now();
        // chunked alternative on ONE thread: 8 chunks each way, interleaved
        const size_t c = n / 8;
        for (int i = 0; i < 8; ++i) {
            hipMemcpyAsync((char *)din + i * c, hin.data() + i * c, c, hipMemcpyHostToDevice, s1);
            hipMemcpyAsync(hout.data() + i * c, (char *)dout + i * c, c, hipMemcpyDeviceToHost, s2);
        }
        hipStreamSynchronize(s1); hipStreamSynchronize(s2);
        double t4 = now();
        printf("H2D %.0f us  D2H %.0f us  two threads both %.0f us  one thread interleaved chunks %.0f us\n",
               (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6);
    }
    return 0;
}
