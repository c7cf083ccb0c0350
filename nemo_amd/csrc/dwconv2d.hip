// Depthwise 3x3 stride-2 convolution on channels-last feature maps: the 'dw_striding' sub-sampling of FastConformer (x8, 256
// channels) and Squeezeformer (x4) -- ConvSubsampling with subsampling='dw_striding',
// nemo/collections/asr/parts/submodules/subsampling.py:142-215 (each further factor of 2 is
// Conv2d(C, C, 3, stride 2, padding 1, groups=C) -> Conv2d(C, C, 1) -> ReLU under MaskedConvSequential :725-759).
//   in  [B, T1, F1, C]  (ReLU'd and time-masked output of the previous stage)
//   out [B, T2, F2, C]  = depthwise(in) + bias,  T2 = (T1 - 1)/2 + 1, F2 = (F1 - 1)/2 + 1
// The pointwise 1x1 convolution that follows is an MFMA GEMM on [B*T2*F2, C] with the ReLU + time-mask epilogue
// (mi355x_gemm, EPI_RELU_MASK), so the mask between the two convolutions needs no pass of its own.
// All three kernels are HBM-bound streams: lane = 8 (bf16) / 4 (f32) consecutive channels = one 16-byte access per tap;
// the nine taps of neighbouring outputs overlap in L2.  Weight / bias gradients: per-workgroup partial sums, then the
// shared second-stage reduction (tap_reduce_kernel, common.h) -- no same-address atomics.
#include "common.h"
#include "mi355x_asr.h"

#define DISPATCH_DT(dt, T, ...)                                      \
  if ((dt) == MI_DT_F32) { typedef float T; __VA_ARGS__; }           \
  else { typedef bf16_t T; __VA_ARGS__; }

// IDX = the type positions are counted and decomposed in: uint32_t whenever B*T*F fits (always, in practice; 64-bit `%` and `/`
// are emulated).  Measured in round 6: no change (input-gradient kernel 848 -> 856 us on Squeezeformer-Medium's first stage) -- the
// kernels are bound by their traffic, which is larger than the header suggests: the input gradient also reads the forward INPUT
// (840 MB, for the ReLU mask) next to writing 840 MB, and the weight-gradient kernel reads it again.  Next step if it matters:
// both gradients from one pass over the input positions (the taps' dout values are the same loads).
template <typename TT, typename IDX>
__global__ __launch_bounds__(256) void dwconv2d_s2_fwd_kernel(const TT* __restrict__ in, const float* __restrict__ w,
                                                              const float* __restrict__ bias, TT* __restrict__ out, int B, int T1,
                                                              int F1, int T2, int F2, int C, int pad) {
  constexpr int V = VecIO<TT>::V;
  const int CP = C / V;  // channel chunks; a thread keeps ONE chunk (its 9 x V weights stay in registers)
  const int ck = threadIdx.x % CP, ps = threadIdx.x / CP, PS = 256 / CP;
  if (ps >= PS) return;
  const int c = ck * V;
  float wk[9][V], bs[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    bs[j] = bias[c + j];
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k][j] = w[(c + j) * 9 + k];
  }
  const IDX npos = (IDX)B * (IDX)T2 * (IDX)F2;
  for (IDX pos = (IDX)blockIdx.x * PS + ps; pos < npos; pos += (IDX)gridDim.x * PS) {
    const int f2 = (int)(pos % (IDX)F2);
    const IDX bt = pos / (IDX)F2;
    const int t2 = (int)(bt % (IDX)T2), b = (int)(bt / (IDX)T2);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = bs[j];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int t1 = 2 * t2 - pad + kh;
      if ((unsigned)t1 >= (unsigned)T1) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int f1 = 2 * f2 - pad + kw;
        if ((unsigned)f1 >= (unsigned)F1) continue;
        float x[V];
        VecIO<TT>::load(in + (((long long)b * T1 + t1) * F1 + f1) * C + c, x);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = fmaf(wk[kh * 3 + kw][j], x[j], acc[j]);
      }
    }
    VecIO<TT>::store(out + (long long)pos * C + c, acc);
  }
}

// din[b,t1,f1,c] = (in > 0) * sum over the outputs (t2, f2) that read (t1, f1):  t1 = 2 t2 - 1 + kh  <=>  kh = t1 + 1 - 2 t2
template <typename TT, typename IDX>
__global__ __launch_bounds__(256) void dwconv2d_s2_bwd_data_kernel(const TT* __restrict__ dout, const TT* __restrict__ in,
                                                                   const float* __restrict__ w, TT* __restrict__ din, int B, int T1,
                                                                   int F1, int T2, int F2, int C, int pad) {
  constexpr int V = VecIO<TT>::V;
  const int CP = C / V;
  const int ck = threadIdx.x % CP, ps = threadIdx.x / CP, PS = 256 / CP;
  if (ps >= PS) return;
  const int c = ck * V;
  float wk[9][V];
#pragma unroll
  for (int j = 0; j < V; ++j)
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k][j] = w[(c + j) * 9 + k];
  const IDX npos = (IDX)B * (IDX)T1 * (IDX)F1;
  for (IDX pos = (IDX)blockIdx.x * PS + ps; pos < npos; pos += (IDX)gridDim.x * PS) {
    const int f1 = (int)(pos % (IDX)F1);
    const IDX bt = pos / (IDX)F1;
    const int t1 = (int)(bt % (IDX)T1), b = (int)(bt / (IDX)T1);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    // kh = t1 + 1 - 2 t2 in {0,1,2}: an even t1 is read through kh = 1 only, an odd one through kh = 0 (t2 = (t1+1)/2) and 2
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int tt = t1 + pad - kh;
      if (tt < 0 || (tt & 1)) continue;
      const int t2 = tt >> 1;
      if (t2 >= T2) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ff = f1 + pad - kw;
        if (ff < 0 || (ff & 1)) continue;
        const int f2 = ff >> 1;
        if (f2 >= F2) continue;
        float g[V];
        VecIO<TT>::load(dout + (((long long)b * T2 + t2) * F2 + f2) * C + c, g);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = fmaf(wk[kh * 3 + kw][j], g[j], acc[j]);
      }
    }
    float x[V];
    VecIO<TT>::load(in + (long long)pos * C + c, x);
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = x[j] > 0.f ? acc[j] : 0.f;  // ReLU of the previous stage
    VecIO<TT>::store(din + (long long)pos * C + c, acc);
  }
}

// partial[part][k][c] (k < 9: weight taps, k = 9: bias) over a contiguous range of output positions per workgroup
template <typename TT, typename IDX>
__global__ __launch_bounds__(256) void dwconv2d_s2_bwd_w_kernel(const TT* __restrict__ dout, const TT* __restrict__ in,
                                                                float* __restrict__ partial, int B, int T1, int F1, int T2, int F2,
                                                                int C, int pos_per_block, int pad) {
  constexpr int V = VecIO<TT>::V;
  extern __shared__ float red[];  // [PS][10][CP*V] reduction over the position slots of the block
  const int CP = C / V;
  const int ck = threadIdx.x % CP, ps = threadIdx.x / CP, PS = 256 / CP;
  const int c = ck * V;
  float gw[10][V];
#pragma unroll
  for (int k = 0; k < 10; ++k)
#pragma unroll
    for (int j = 0; j < V; ++j) gw[k][j] = 0.f;
  const IDX npos = (IDX)B * (IDX)T2 * (IDX)F2;
  const IDX p0 = (IDX)blockIdx.x * (IDX)pos_per_block, p1 = min(npos, p0 + (IDX)pos_per_block);
  if (ps < PS) {
    for (IDX pos = p0 + ps; pos < p1; pos += PS) {
      const int f2 = (int)(pos % (IDX)F2);
      const IDX bt = pos / (IDX)F2;
      const int t2 = (int)(bt % (IDX)T2), b = (int)(bt / (IDX)T2);
      float g[V];
      VecIO<TT>::load(dout + (long long)pos * C + c, g);
#pragma unroll
      for (int j = 0; j < V; ++j) gw[9][j] += g[j];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int t1 = 2 * t2 - pad + kh;
        if ((unsigned)t1 >= (unsigned)T1) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int f1 = 2 * f2 - pad + kw;
          if ((unsigned)f1 >= (unsigned)F1) continue;
          float x[V];
          VecIO<TT>::load(in + (((long long)b * T1 + t1) * F1 + f1) * C + c, x);
#pragma unroll
          for (int j = 0; j < V; ++j) gw[kh * 3 + kw][j] = fmaf(g[j], x[j], gw[kh * 3 + kw][j]);
        }
      }
    }
  }
  // reduce over the PS position slots through LDS, one tap per round (unrolled: gw[] must stay in registers)
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    __syncthreads();
    if (ps < PS) {
#pragma unroll
      for (int j = 0; j < V; ++j) red[ps * C + c + j] = gw[k][j];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < C; e += 256) {
      float s = 0.f;
      for (int q = 0; q < PS; ++q) s += red[q * C + e];
      partial[((long long)blockIdx.x * 10 + k) * C + e] = s;
    }
  }
}

static inline int dw2d_grid(long long npos, int PS) {
  long long g = (npos + PS - 1) / PS;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

extern "C" int mi355x_dwconv2d_s2_fwd_pad(const void* in, const void* w, const void* bias, void* out, int dt, int B, int T1, int F1,
                                          int C, int pad, void* stream);
extern "C" int mi355x_dwconv2d_s2_fwd(const void* in, const void* w, const void* bias, void* out, int dt, int B, int T1, int F1,
                                      int C, void* stream) {
  return mi355x_dwconv2d_s2_fwd_pad(in, w, bias, out, dt, B, T1, F1, C, 1, stream);
}
// pad = zero rows / columns in front of the grid (1: Conv2d(padding = 1); 2: CausalConv2D, causal_convs.py:24-72); one behind it
extern "C" int mi355x_dwconv2d_s2_fwd_pad(const void* in, const void* w, const void* bias, void* out, int dt, int B, int T1, int F1,
                                          int C, int pad, void* stream) {
  mi_clear_errors();
  const int V = dt == MI_DT_BF16 ? 8 : 4;
  if (!in || !w || !bias || !out || B <= 0 || T1 <= 0 || F1 <= 0 || C <= 0 || C % V || C / V > 256 || pad < 1 || pad > 2) return MI_ERR_ARG;
  const int T2 = (T1 + pad - 2) / 2 + 1, F2 = (F1 + pad - 2) / 2 + 1;
  const int PS = 256 / (C / V);
  // (positions + one grid stride must fit the index type)
  const bool small = (long long)B * T1 * F1 + 8192LL * 256 < (1LL << 32);
#define DW2D_FWD(IDX) DISPATCH_DT(dt, TT, MI_LAUNCH((dwconv2d_s2_fwd_kernel<TT, IDX>), dim3(dw2d_grid((long long)B * T2 * F2, PS)), dim3(256), 0, \
                                         (hipStream_t)stream, (const TT*)in, (const float*)w, (const float*)bias, (TT*)out, B, T1, F1, \
                                         T2, F2, C, pad))
  if (small) { DW2D_FWD(uint32_t); } else { DW2D_FWD(long long); }
#undef DW2D_FWD
  return mi_check_launch();
}

extern "C" int mi355x_dwconv2d_s2_bwd_pad(const void* dout, const void* in, const void* w, void* din, void* dw, void* dbias, int dt,
                                          int B, int T1, int F1, int C, int pad, void* scratch, long long scratch_elems, void* stream);
extern "C" int mi355x_dwconv2d_s2_bwd(const void* dout, const void* in, const void* w, void* din, void* dw, void* dbias, int dt,
                                      int B, int T1, int F1, int C, void* scratch, long long scratch_elems, void* stream) {
  return mi355x_dwconv2d_s2_bwd_pad(dout, in, w, din, dw, dbias, dt, B, T1, F1, C, 1, scratch, scratch_elems, stream);
}
extern "C" int mi355x_dwconv2d_s2_bwd_pad(const void* dout, const void* in, const void* w, void* din, void* dw, void* dbias, int dt,
                                          int B, int T1, int F1, int C, int pad, void* scratch, long long scratch_elems, void* stream) {
  mi_clear_errors();
  const int V = dt == MI_DT_BF16 ? 8 : 4;
  if (!dout || !in || !w || !din || !dw || !dbias || !scratch || B <= 0 || T1 <= 0 || F1 <= 0 || C <= 0 || C % V || C / V > 256 ||
      pad < 1 || pad > 2)
    return MI_ERR_ARG;
  const int T2 = (T1 + pad - 2) / 2 + 1, F2 = (F1 + pad - 2) / 2 + 1;
  const int PS = 256 / (C / V);
  const long long npos = (long long)B * T2 * F2;
  const long long nblk = npos < 1024 * 64 ? (npos + 63) / 64 : 1024;
  if (scratch_elems < nblk * 10 * C) return MI_ERR_ARG;
  const int per = (int)((npos + nblk - 1) / nblk);
  hipStream_t s = (hipStream_t)stream;
  const size_t shm = (size_t)PS * C * sizeof(float);
  if (shm > 64 * 1024) return MI_ERR_ARG;
  const bool small = (long long)B * T1 * F1 + 8192LL * 256 < (1LL << 32);
#define DW2D_BWD(IDX) do { \
  DISPATCH_DT(dt, TT, MI_LAUNCH((dwconv2d_s2_bwd_data_kernel<TT, IDX>), dim3(dw2d_grid((long long)B * T1 * F1, PS)), dim3(256), 0, \
                                         s, (const TT*)dout, (const TT*)in, (const float*)w, (TT*)din, B, T1, F1, T2, F2, C, pad)); \
  DISPATCH_DT(dt, TT, MI_LAUNCH((dwconv2d_s2_bwd_w_kernel<TT, IDX>), dim3((unsigned)nblk), dim3(256), shm, s, (const TT*)dout, \
                                         (const TT*)in, (float*)scratch, B, T1, F1, T2, F2, C, per, pad)); } while (0)
  if (small) DW2D_BWD(uint32_t); else DW2D_BWD(long long);
#undef DW2D_BWD
  MI_LAUNCH(tap_reduce_kernel, dim3((10 * C + 255) / 256, 16), dim3(256), 0, s, (const float*)scratch, (int)nblk, 9, C,
                     (float*)dw, (float*)dbias);
  return mi_check_launch();
}
