// occupancy of a 384-thread, 72-VGPR-class kernel as a function of dynamic LDS (gfx950: 160 KiB/CU)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(384) k(float *p) { extern __shared__ float s[]; s[threadIdx.x] = p[threadIdx.x]; __syncthreads(); p[threadIdx.x] = s[threadIdx.x ^ 1]; }
__global__ void __launch_bounds__(256) k256(float *p) { extern __shared__ float s[]; s[threadIdx.x] = p[threadIdx.x]; __syncthreads(); p[threadIdx.x] = s[threadIdx.x ^ 1]; }
int main() {
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipFuncSetAttribute((const void *)k256, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    for (int lds : {20480, 32768, 38400, 40000, 40448, 40960, 41472, 42240, 54000, 54272, 54784, 65536, 81920, 82000, 163840}) {
        int n = -1, m = -1;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 384, lds);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&m, k256, 256, lds);
        printf("lds %6d: %d blocks/CU (384 thr)  %d (256 thr)\n", lds, n, m);
    }
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    printf("sharedMemPerMultiprocessor %zu maxSharedMemoryPerMultiProcessor %zu sharedMemPerBlock %zu\n", (size_t)pr.sharedMemPerMultiprocessor, (size_t)pr.maxSharedMemoryPerMultiProcessor, (size_t)pr.sharedMemPerBlock);
}
