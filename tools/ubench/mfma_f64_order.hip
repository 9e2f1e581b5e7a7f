// mfma_f64_order — is v_mfma_f64_16x16x4_f64 the k-ordered fma chain (as the f32-input form is)?  The canonical order
// of the exact engine is  acc = fma(a_k, b_k, acc)  for ascending k; if the hardware evaluates
//   D = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, C))))
// bit for bit, the float64 engine (float64 / int32 I/O) can move from the vector ALU to the matrix pipe.
// Candidates tested per element over many random operand sets (bitwise equality):
//   0 ascending fma chain   1 descending fma chain   2 exact products summed pairwise   3 two-term tree of fmas
// Also checks the C/D layout (row = (lane >> 4) + 4 * reg).   Build: hipcc --offload-arch=gfx950 -O3 mfma_f64_order.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k(const double *A, const double *B, const double *C, double *D, int chain)
{
    // A[16][4], B[4][16], C[16][16] row-major; one wave.  `chain` MFMAs in a row, each with its own A/B slab.
    const int lane = threadIdx.x, r = lane & 15, kk = lane >> 4;
    f64x4 acc;
    for (int v = 0; v < 4; ++v) acc[v] = C[((lane >> 4) + 4 * v) * 16 + (lane & 15)];
    for (int s = 0; s < chain; ++s)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[s * 64 + r * 4 + kk], B[s * 64 + kk * 16 + r], acc, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[((lane >> 4) + 4 * v) * 16 + (lane & 15)] = acc[v];
}

int main()
{
    const int chain = 8, trials = 200;
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd(0., 1.);
    double *dA, *dB, *dC, *dD;
    CHECK(hipMalloc(&dA, chain * 64 * 8)); CHECK(hipMalloc(&dB, chain * 64 * 8)); CHECK(hipMalloc(&dC, 256 * 8)); CHECK(hipMalloc(&dD, 256 * 8));
    long match[4] = {0, 0, 0, 0}, total = 0;
    for (int t = 0; t < trials; ++t) {
        std::vector<double> A(chain * 64), B(chain * 64), C(256), D(256);
        for (auto &v : A) v = nd(rng) * std::exp2((double)(rng() % 24) - 12.);
        for (auto &v : B) v = nd(rng);
        for (auto &v : C) v = nd(rng) * 1e-3;
        CHECK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, nullptr, dA, dB, dC, dD, chain);
        CHECK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double c0 = C[i * 16 + j], c1 = c0, c2 = c0, c3 = c0;
                for (int s = 0; s < chain; ++s) {
                    const double *a = &A[s * 64 + i * 4];
                    double b[4];
                    for (int q = 0; q < 4; ++q) b[q] = B[s * 64 + q * 16 + j];
                    for (int q = 0; q < 4; ++q) c0 = std::fma(a[q], b[q], c0);
                    for (int q = 3; q >= 0; --q) c1 = std::fma(a[q], b[q], c1);
                    c2 = c2 + ((a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]));
                    c3 = std::fma(a[3], b[3], std::fma(a[2], b[2], 0.)) + std::fma(a[1], b[1], std::fma(a[0], b[0], c3));
                }
                const double d = D[i * 16 + j];
                match[0] += d == c0; match[1] += d == c1; match[2] += d == c2; match[3] += d == c3;
                ++total;
            }
    }
    printf("v_mfma_f64_16x16x4_f64, %d chained MFMAs, %ld elements:\n", chain, total);
    const char *names[] = {"ascending fma chain (canonical order)", "descending fma chain", "products summed pairwise", "two-term tree"};
    for (int c = 0; c < 4; ++c) printf("  %-40s bitwise equal in %ld / %ld\n", names[c], match[c], total);
    return 0;
}
