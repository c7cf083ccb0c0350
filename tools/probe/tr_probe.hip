// probe of ds_read_b64_tr_b16 semantics on gfx950: prints, per lane, which (row, col) of the LDS image each of the 4
// returned b16 elements came from, when lane t of each 16-lane group supplies &img[(t/4) + 4*g][(t%4)*4].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(uint16_t* out, int stride) {
  __shared__ __attribute__((aligned(16))) uint16_t sm[64 * 128];
  for (int i = threadIdx.x; i < 64 * 128; i += 64) sm[i] = (uint16_t)i;
  __syncthreads();
  const int t = threadIdx.x & 15, g = threadIdx.x >> 4;
  const uint16_t* addr = sm + ((t >> 2) + 4 * g) * stride + (t & 3) * 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  const int stride = 128;
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, stride);
  uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) printf(" (r%2d,c%2d)", h[l * 4 + j] / stride, h[l * 4 + j] % stride);
    printf("\n");
  }
  return 0;
}
