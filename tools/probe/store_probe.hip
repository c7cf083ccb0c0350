// store-path probe: how fast can 256x128 bf16 output tiles be written, as a function of pattern and occupancy?
//   hipcc --offload-arch=gfx950 -O3 store_probe.hip -o store_probe && ./store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
// mode 0: GEMM-epilogue pattern (thread -> row tid>>4 + 32*it, 16 B at (tid&15)*16 B), tiles mapped m-major like the GEMM
// mode 1: same tile, but one wave writes 4 KB = 16 full... (wave -> 16 rows x 256 B, it over row groups)   [same as 0, kept for A/B]
// mode 2: contiguous 64 KiB per block (fill-like)
// mode 3: tile pattern with 512-B row segments (256x256 tile, 2 B elems): thread -> row tid>>5 + 16*it, 16 B at (tid&31)*16
template <int MODE>
__global__ __launch_bounds__(512) void k(uint16_t* C, int M, int N, int tn, int ntiles) {
  extern __shared__ char lds[];
  const int bid = blockIdx.x;
  const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  u32x4 v = {(uint32_t)threadIdx.x, (uint32_t)bid, 3u, 4u};
  if (MODE == 0 || MODE == 1) {
    const int tile_m = logical / tn, tile_n = logical - tile_m * tn;
    const int m0 = tile_m * 256, n0 = tile_n * 128;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int m = m0 + (threadIdx.x >> 4) + 32 * it, n = n0 + (threadIdx.x & 15) * 8;
      if (m < M) *reinterpret_cast<u32x4*>(C + (long long)m * N + n) = v;
    }
  } else if (MODE == 2) {
    uint16_t* base = C + (long long)logical * 256 * 128;
#pragma unroll
    for (int it = 0; it < 8; ++it) *reinterpret_cast<u32x4*>(base + (it * 512 + threadIdx.x) * 8) = v;
  } else {
    const int tn2 = tn / 2;
    const int tile_m = logical / tn2, tile_n = logical - tile_m * tn2;
    const int m0 = tile_m * 256, n0 = tile_n * 256;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int m = m0 + (threadIdx.x >> 5) + 16 * it, n = n0 + (threadIdx.x & 31) * 8;
      if (m < M) *reinterpret_cast<u32x4*>(C + (long long)m * N + n) = v;
    }
  }
}
template <int MODE>
void run(const char* name, uint16_t* C, int M, int N, int shm) {
  const int tm = (M + 255) / 256, tn = N / 128;
  int ntiles = tm * tn;
  if (MODE == 3) ntiles /= 2;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(ntiles), dim3(512), shm, 0, C, M, N, tn, ntiles);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<MODE>, dim3(ntiles), dim3(512), shm, 0, C, M, N, tn, ntiles);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 20, bytes = (double)M * N * 2;
  printf("%-44s M=%d N=%d lds=%3d KiB  %7.1f us  %5.2f TB/s\n", name, M, N, shm / 1024, us, bytes / us / 1e6);
}
int main() {
  uint16_t* C; hipMalloc(&C, 512ll << 20);
  for (int N : {2048, 512}) {
    const int M = 16032 * (2048 / N);
    for (int shm : {144 * 1024, 64 * 1024, 16 * 1024}) {
      run<0>("tile 256x128, 256-B row segments", C, M, N, shm);
      run<3>("tile 256x256, 512-B row segments", C, M, N, shm);
      run<2>("contiguous 64 KiB per block", C, M, N, shm);
    }
  }
  return 0;
}
