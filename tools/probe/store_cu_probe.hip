// per-CU store rate: W workgroups (one per CU: 128 KiB of LDS each), each writing T 256x256 bf16 tiles (512-B row segments, 16 B per
// lane) -- is a CU's store path limited by itself (time independent of W) or by the fabric (time ~ W)?
//   hipcc --offload-arch=gfx950 -O3 store_cu_probe.hip -o store_cu_probe && ./store_cu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__global__ __launch_bounds__(512) void k(uint16_t* C, int N, int T, int stride_tiles) {
  extern __shared__ char lds[];
  u32x4 v = {(uint32_t)threadIdx.x, (uint32_t)blockIdx.x, 3u, 4u};
  for (int t = 0; t < T; ++t) {
    const long long tile = (long long)blockIdx.x * stride_tiles + t;
    const long long tn = N / 256;
    const long long m0 = (tile / tn) * 256, n0 = (tile % tn) * 256;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const long long m = m0 + (threadIdx.x >> 5) + 16 * it, n = n0 + (threadIdx.x & 31) * 8;
      *reinterpret_cast<u32x4*>(C + m * N + n) = v;
    }
  }
}
int main() {
  uint16_t* C; hipMalloc(&C, 1024ll << 20);
  const int shm = 128 * 1024, N = 2048;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int T : {1, 2, 8}) for (int W : {8, 32, 64, 128, 256}) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(W), dim3(512), shm, 0, C, N, T, 8);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(W), dim3(512), shm, 0, C, N, T, 8);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 20, bytes = (double)W * T * 131072;
    printf("W=%3d workgroups x T=%d tiles of 128 KiB: %7.2f us per launch  %6.3f TB/s  %6.2f GB/s per CU\n", W, T, us, bytes / us / 1e6, bytes / us / 1e3 / W);
  }
  return 0;
}
