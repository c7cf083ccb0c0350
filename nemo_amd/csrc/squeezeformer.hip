// HBM-bound glue kernels of the Squeezeformer block (BASELINE.json configs[4]).  The dense work
// (q|k|v, FFN, pointwise convolutions, time-recovery Linear and every gradient GEMM) is mi355x_gemm; the depthwise
// k=31 convolution / BatchNorm / Swish of the convolution module are the Conformer kernels of convmod.hip run on 2*d_model
// channels.  What the post-LN block adds on top, replacing on the reference path:
//   ScaleBiasLayer (y = x*scale + bias ahead of every sub-block)         squeezeformer_modules.py:30-57, :143,:166,:172,:178
//   Swish + pad mask after pointwise_conv1 (instead of GLU)               conformer_modules.py:267-275, 324-331
//   TimeReductionModule's masked depthwise Conv1d(k=5, s=2, pad=3)        subsampling.py:589-646
//   time recovery: repeat_interleave(2)[:, :T] of the Linear's output + skip   squeezeformer_encoder.py:352-361
// Row pitches: bf16 GEMM operands must start every row on a 16-byte boundary, and Squeezeformer's d_model is not always a
// multiple of 8 (Medium: 324).  Every kernel here that PRODUCES a K-contiguous GEMM operand therefore takes an explicit
// output pitch `ld` >= d (a multiple of 4) and zero-fills the columns [d, ld).
#include "common.h"
#include "mi355x_asr.h"

#define DISPATCH_DT(dt, T, ...)                                      \
  if ((dt) == MI_DT_F32) { typedef float T; __VA_ARGS__; }           \
  else { typedef bf16_t T; __VA_ARGS__; }

static inline int sgrid(long long n) { long long g = (n + 255) / 256; return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); }

// ------------------------------------------------------------------------------------------------ ScaleBias
template <typename TY>
__global__ __launch_bounds__(256) void scale_bias_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ bias, TY* __restrict__ y, long long M,
                                                             int d, int ld) {
  const int lv = ld >> 2;
  const long long nv = M * lv;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    const long long m = i / lv;
    const int c = (int)(i - m * lv) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < d) {
      ld4(x + m * d + c, v);
      if (scale) {
        float s[4], b[4];
        ld4(scale + c, s); ld4(bias + c, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] * s[j] + b[j];
      }
    }
    st4(y + m * ld + c, v);
  }
}

// dres[m, c] += dy[m, c] * scale[c];  dscale[c] += sum_m dy[m, c] * x[m, c];  dbias[c] += sum_m dy[m, c]
// A block owns a contiguous range of rows; its 256 threads are (column group of 4) x (row lane); the row lanes' partial
// column sums meet in LDS and go out as ONE plain store per column into the block's slab partial[blk][2][d]; the slabs are
// summed by sb_reduce_kernel (same-address float atomics from 512 blocks serialise at the memory side: 65 us per launch
// at M = 16032, d = 324 against 15 us of streaming).
template <typename TDY>
__global__ __launch_bounds__(256) void scale_bias_bwd_kernel(const TDY* __restrict__ dy, int ld, const float* __restrict__ x,
                                                             const float* __restrict__ scale, float* __restrict__ dres,
                                                             float* __restrict__ partial, long long M, int d, int rows_per_block) {
  __shared__ float red[256 * 8];
  const int ncg = d >> 2;
  const int W = ncg < 256 ? ncg : 256;   // column groups handled at once
  const int R = 256 / W;                 // row lanes
  const int cgl = threadIdx.x % W, rl = threadIdx.x / W;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  float* slab = partial ? partial + (long long)blockIdx.x * 2 * d : nullptr;
  for (int cg0 = 0; cg0 < ncg; cg0 += W) {
    const int cg = cg0 + cgl;
    const bool on = cg < ncg && rl < R;
    const int c = cg * 4;
    float as[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
    if (on) {
      float s[4] = {1.f, 1.f, 1.f, 1.f};
      if (scale) ld4(scale + c, s);
      for (long long m = r0 + rl; m < r1; m += R) {
        float g[4], xv[4], r[4];
        ld4(dy + m * ld + c, g); ld4(x + m * d + c, xv); ld4(dres + m * d + c, r);
#pragma unroll
        for (int j = 0; j < 4; ++j) { r[j] += g[j] * s[j]; as[j] += g[j] * xv[j]; ab[j] += g[j]; }
        st4(dres + m * d + c, r);
      }
    }
    if (slab) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { red[threadIdx.x * 8 + j] = as[j]; red[threadIdx.x * 8 + 4 + j] = ab[j]; }
      __syncthreads();
      if (on && rl == 0) {
        float ss[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < R; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) { ss[j] += red[(q * W + cgl) * 8 + j]; sb[j] += red[(q * W + cgl) * 8 + 4 + j]; }
        st4(slab + c, ss); st4(slab + d + c, sb);
      }
      __syncthreads();
    }
  }
}
// dscale[c] += sum_p partial[p][0][c];  dbias[c] += sum_p partial[p][1][c];  grid (ceil(2d/256), nsplit)
static __global__ __launch_bounds__(256) void sb_reduce_kernel(const float* __restrict__ partial, int nparts, int d,
                                                               float* __restrict__ dscale, float* __restrict__ dbias) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * d) return;
  const int per = (nparts + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * per, p1 = min(nparts, p0 + per);
  float a0 = 0.f, a1 = 0.f;
  int p = p0;
  for (; p + 2 <= p1; p += 2) { a0 += partial[(long long)p * 2 * d + i]; a1 += partial[(long long)(p + 1) * 2 * d + i]; }
  for (; p < p1; ++p) a0 += partial[(long long)p * 2 * d + i];
  atomicAdd(i < d ? dscale + i : dbias + (i - d), a0 + a1);
}

// y[m, 0:d] = alpha * dropmask(m*d + c) * x[m, c]  (pitch ld, zero pad): the residual-branch gradient as a GEMM operand
template <typename TY>
__global__ __launch_bounds__(256) void cast_pitched_kernel(const float* __restrict__ x, TY* __restrict__ y, long long M, int d, int ld,
                                                           float alpha, DropCfg drop) {
  drop_resolve(drop);
  const int lv = ld >> 2;
  const long long nv = M * lv;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    const long long m = i / lv;
    const int c = (int)(i - m * lv) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < d) {
      ld4(x + m * d + c, v);
      const uint32_t idx = (uint32_t)(m * d + c);
      float dm[8];
      drop_mask8(drop, idx & ~7u, dm);
      const int h = (idx & 4u) ? 4 : 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] *= alpha * dm[h + j];
    }
    st4(y + m * ld + c, v);
  }
}

// ------------------------------------------------------------------------------------------------ Swish + pad mask
template <typename T>
__global__ __launch_bounds__(256) void swish_mask_fwd_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                             const long long* __restrict__ len, int Tt, long long M, int C) {
  const int cv = C >> 2;
  const long long nv = M * cv;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    const long long m = i / cv;
    const int b = (int)(m / Tt), t = (int)(m - (long long)b * Tt);
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (!len || t < len[b]) {
      ld4(in + i * 4, o);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = swishf_(o[j]);
    }
    st4(out + i * 4, o);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void swish_mask_bwd_kernel(const T* __restrict__ in, const T* __restrict__ dout, T* __restrict__ din,
                                                             const long long* __restrict__ len, int Tt, long long M, int C) {
  const int cv = C >> 2;
  const long long nv = M * cv;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    const long long m = i / cv;
    const int b = (int)(m / Tt), t = (int)(m - (long long)b * Tt);
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (!len || t < len[b]) {
      float a[4];
      ld4(in + i * 4, a); ld4(dout + i * 4, o);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] *= swish_grad(a[j]);
    }
    st4(din + i * 4, o);
  }
}

// ------------------------------------------------------------------------------------------------ time reduction (dw k=5 s=2)
// out[b, t', c] = bias[c] + sum_k w[c, k] * xm[b, 2t' - 3 + k, c],  xm = x masked to t < len[b] and zero outside [0, T);
// t' in [0, Th), Th = ceil(T/2) (the convolution yields floor((T+1)/2) + 1 frames; the reference crops to Th).
template <typename TY>
__global__ __launch_bounds__(256) void tr_dwconv_fwd_kernel(const float* __restrict__ x, const long long* __restrict__ len,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            TY* __restrict__ out, int B, int T, int Th, int d, int ld) {
  const int lv = ld >> 2;
  const long long nv = (long long)B * Th * lv;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    const long long r = i / lv;
    const int c = (int)(i - r * lv) * 4;
    const int b = (int)(r / Th), tp = (int)(r - (long long)b * Th);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < d) {
      ld4(bias + c, acc);
      const int L = len ? (int)(len[b] < T ? len[b] : T) : T;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int t = 2 * tp - 3 + k;
        if (t >= 0 && t < L) {
          float xv[4];
          ld4(x + ((long long)b * T + t) * d + c, xv);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] += w[(c + j) * 5 + k] * xv[j];
        }
      }
    }
    st4(out + r * ld + c, acc);
  }
}
// dx[b, t, c] += [t < len[b]] * sum_{k == (t+3) mod 2} w[c, k] * dout[b, (t + 3 - k)/2, c]
template <typename TY>
__global__ __launch_bounds__(256) void tr_dwconv_bwd_data_kernel(const TY* __restrict__ dout, int ld, const long long* __restrict__ len,
                                                                 const float* __restrict__ w, float* __restrict__ dx, int B, int T,
                                                                 int Th, int d) {
  const int dv = d >> 2;
  const long long nv = (long long)B * T * dv;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    const long long r = i / dv;
    const int c = (int)(i - r * dv) * 4;
    const int b = (int)(r / T), t = (int)(r - (long long)b * T);
    if (len && t >= len[b]) continue;
    float acc[4];
    ld4(dx + r * d + c, acc);
    for (int k = (t + 3) & 1; k < 5; k += 2) {
      const int tp = (t + 3 - k) >> 1;
      if (tp >= 0 && tp < Th) {
        float g[4];
        ld4(dout + ((long long)b * Th + tp) * ld + c, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += w[(c + j) * 5 + k] * g[j];
      }
    }
    st4(dx + r * d + c, acc);
  }
}
// dw[c, k] += sum_{b,t'} dout[b,t',c] * xm[b, 2t'-3+k, c];  dbias[c] += sum dout[b,t',c]     (rows r = b*Th + t')
template <typename TY>
__global__ __launch_bounds__(256) void tr_dwconv_bwd_w_kernel(const TY* __restrict__ dout, int pitch, const float* __restrict__ x,
                                                              const long long* __restrict__ len, float* __restrict__ dw,
                                                              float* __restrict__ dbias, int B, int T, int Th, int d,
                                                              int rows_per_block) {
  __shared__ float red[256 * 6];
  const int W = d < 256 ? d : 256;
  const int R = 256 / W;
  const int cl = threadIdx.x % W, rl = threadIdx.x / W;
  const long long Mh = (long long)B * Th;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = r0 + rows_per_block < Mh ? r0 + rows_per_block : Mh;
  for (int c0 = 0; c0 < d; c0 += W) {
    const int c = c0 + cl;
    const bool on = c < d && rl < R;
    float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (on) {
      for (long long r = r0 + rl; r < r1; r += R) {
        const int b = (int)(r / Th), tp = (int)(r - (long long)b * Th);
        const int L = len ? (int)(len[b] < T ? len[b] : T) : T;
        const float g = ld(dout + r * pitch + c);
        a[5] += g;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          const int t = 2 * tp - 3 + k;
          if (t >= 0 && t < L) a[k] += g * x[((long long)b * T + t) * d + c];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) red[threadIdx.x * 6 + k] = a[k];
    __syncthreads();
    if (on && rl == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        float s = 0.f;
        for (int q = 0; q < R; ++q) s += red[(q * W + cl) * 6 + k];
        if (k < 5) atomicAdd(dw + c * 5 + k, s);
        else atomicAdd(dbias + c, s);
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ time recovery
// out[b, t, :] = skip[b, t, :] + ys[b, t >> 1, :]      (repeat_interleave(2)[:, :T] of the Linear's output + the cached input)
__global__ __launch_bounds__(256) void time_recover_fwd_kernel(const float* __restrict__ skip, const float* __restrict__ ys,
                                                               float* __restrict__ out, int B, int T, int Th, int d) {
  const int dv = d >> 2;
  const long long nv = (long long)B * T * dv;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    const long long r = i / dv;
    const int c = (int)(i - r * dv) * 4;
    const int b = (int)(r / T), t = (int)(r - (long long)b * T);
    float a[4], y[4];
    ld4(skip + r * d + c, a); ld4(ys + ((long long)b * Th + (t >> 1)) * d + c, y);
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] += y[j];
    st4(out + r * d + c, a);
  }
}
// dys[b, t', :] = dx[b, 2t', :] + dx[b, 2t'+1, :] (second term only when 2t'+1 < T); pitch ld, zero pad
template <typename TY>
__global__ __launch_bounds__(256) void time_recover_bwd_kernel(const float* __restrict__ dx, TY* __restrict__ dys, int B, int T, int Th,
                                                               int d, int ld) {
  const int lv = ld >> 2;
  const long long nv = (long long)B * Th * lv;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    const long long r = i / lv;
    const int c = (int)(i - r * lv) * 4;
    const int b = (int)(r / Th), tp = (int)(r - (long long)b * Th);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < d) {
      ld4(dx + ((long long)b * T + 2 * tp) * d + c, a);
      if (2 * tp + 1 < T) {
        float e[4];
        ld4(dx + ((long long)b * T + 2 * tp + 1) * d + c, e);
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] += e[j];
      }
    }
    st4(dys + r * ld + c, a);
  }
}

// =================================================================================================
static inline bool bad_pitch(int d, int ld) { return d <= 0 || (d & 3) || ld < d || (ld & 3); }
static inline int reduce_blocks(long long rows, int* rows_per_block) {
  int nb = (int)(rows < 512 ? rows : 512);
  if (nb < 1) nb = 1;
  *rows_per_block = (int)((rows + nb - 1) / nb);
  return (int)((rows + *rows_per_block - 1) / *rows_per_block);
}

extern "C" int mi355x_scale_bias_fwd(const void* x, const void* scale, const void* bias, void* y, int y_dtype, long long M, int d,
                                     int ld, void* stream) {
  mi_clear_errors();
  if (!x || !y || M <= 0 || bad_pitch(d, ld) || (!scale) != (!bias)) return MI_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_DT(y_dtype, TY, MI_LAUNCH((scale_bias_fwd_kernel<TY>), dim3(sgrid(M * (ld >> 2))), dim3(256), 0, s,
                                              (const float*)x, (const float*)scale, (const float*)bias, (TY*)y, M, d, ld));
  return mi_check_launch();
}
extern "C" int mi355x_scale_bias_bwd(const void* dy, int dy_dtype, int ld, const void* x, const void* scale, void* dres, void* dscale,
                                     void* dbias, long long M, int d, void* scratch, long long scratch_elems, void* stream) {
  mi_clear_errors();
  if (!dy || !x || !dres || M <= 0 || bad_pitch(d, ld) || (!dscale) != (!dbias)) return MI_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  int rpb;
  const int nb = reduce_blocks(M, &rpb);
  if (dscale && (!scratch || scratch_elems < (long long)nb * 2 * d)) return MI_ERR_ARG;
  DISPATCH_DT(dy_dtype, TD, MI_LAUNCH((scale_bias_bwd_kernel<TD>), dim3(nb), dim3(256), 0, s, (const TD*)dy, ld,
                                               (const float*)x, (const float*)scale, (float*)dres,
                                               dscale ? (float*)scratch : (float*)nullptr, M, d, rpb));
  if (dscale)
    MI_LAUNCH(sb_reduce_kernel, dim3((2 * d + 255) / 256, 8), dim3(256), 0, s, (const float*)scratch, nb, d, (float*)dscale,
                       (float*)dbias);
  return mi_check_launch();
}
extern "C" int mi355x_cast_pitched(const void* x, void* y, int y_dtype, long long M, int d, int ld, float alpha, unsigned drop_key,
                                   unsigned drop_threshold, float drop_scale, void* stream) {
  mi_clear_errors();
  if (!x || !y || M <= 0 || bad_pitch(d, ld) || M * d >= (1LL << 32)) return MI_ERR_ARG;
  DropCfg dc = mi_drop(drop_key, drop_threshold, drop_scale);
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_DT(y_dtype, TY, MI_LAUNCH((cast_pitched_kernel<TY>), dim3(sgrid(M * (ld >> 2))), dim3(256), 0, s,
                                              (const float*)x, (TY*)y, M, d, ld, alpha, dc));
  return mi_check_launch();
}
extern "C" int mi355x_swish_mask_fwd(const void* in, void* out, int dtype, const void* len, int T, long long M, int C, void* stream) {
  mi_clear_errors();
  if (!in || !out || M <= 0 || C <= 0 || (C & 3) || T <= 0) return MI_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_DT(dtype, TT, MI_LAUNCH((swish_mask_fwd_kernel<TT>), dim3(sgrid(M * (C >> 2))), dim3(256), 0, s, (const TT*)in,
                                            (TT*)out, (const long long*)len, T, M, C));
  return mi_check_launch();
}
extern "C" int mi355x_swish_mask_bwd(const void* in, const void* dout, void* din, int dtype, const void* len, int T, long long M, int C,
                                     void* stream) {
  mi_clear_errors();
  if (!in || !dout || !din || M <= 0 || C <= 0 || (C & 3) || T <= 0) return MI_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_DT(dtype, TT, MI_LAUNCH((swish_mask_bwd_kernel<TT>), dim3(sgrid(M * (C >> 2))), dim3(256), 0, s, (const TT*)in,
                                            (const TT*)dout, (TT*)din, (const long long*)len, T, M, C));
  return mi_check_launch();
}
extern "C" int mi355x_time_reduce_dwconv_fwd(const void* x, const void* len, const void* w, const void* bias, void* out, int out_dtype,
                                             int B, int T, int d, int ld, void* stream) {
  mi_clear_errors();
  if (!x || !w || !bias || !out || B <= 0 || T <= 0 || bad_pitch(d, ld)) return MI_ERR_ARG;
  const int Th = (T + 1) / 2;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_DT(out_dtype, TY, MI_LAUNCH((tr_dwconv_fwd_kernel<TY>), dim3(sgrid((long long)B * Th * (ld >> 2))), dim3(256), 0, s,
                                                (const float*)x, (const long long*)len, (const float*)w, (const float*)bias,
                                                (TY*)out, B, T, Th, d, ld));
  return mi_check_launch();
}
extern "C" int mi355x_time_reduce_dwconv_bwd(const void* dout, int dout_dtype, int ld, const void* x, const void* len, const void* w,
                                             void* dx, void* dw, void* dbias, int B, int T, int d, void* stream) {
  mi_clear_errors();
  if (!dout || !x || !w || !dx || !dw || !dbias || B <= 0 || T <= 0 || bad_pitch(d, ld)) return MI_ERR_ARG;
  const int Th = (T + 1) / 2;
  hipStream_t s = (hipStream_t)stream;
  int rpb;
  const int nb = reduce_blocks((long long)B * Th, &rpb);
  DISPATCH_DT(dout_dtype, TY,
              MI_LAUNCH((tr_dwconv_bwd_data_kernel<TY>), dim3(sgrid((long long)B * T * (d >> 2))), dim3(256), 0, s,
                                 (const TY*)dout, ld, (const long long*)len, (const float*)w, (float*)dx, B, T, Th, d);
              MI_LAUNCH((tr_dwconv_bwd_w_kernel<TY>), dim3(nb), dim3(256), 0, s, (const TY*)dout, ld, (const float*)x,
                                 (const long long*)len, (float*)dw, (float*)dbias, B, T, Th, d, rpb));
  return mi_check_launch();
}
extern "C" int mi355x_time_recover_fwd(const void* skip, const void* ys, void* out, int B, int T, int d, void* stream) {
  mi_clear_errors();
  if (!skip || !ys || !out || B <= 0 || T <= 0 || d <= 0 || (d & 3)) return MI_ERR_ARG;
  MI_LAUNCH(time_recover_fwd_kernel, dim3(sgrid((long long)B * T * (d >> 2))), dim3(256), 0, (hipStream_t)stream,
                     (const float*)skip, (const float*)ys, (float*)out, B, T, (T + 1) / 2, d);
  return mi_check_launch();
}
extern "C" int mi355x_time_recover_bwd(const void* dx, void* dys, int dys_dtype, int B, int T, int d, int ld, void* stream) {
  mi_clear_errors();
  if (!dx || !dys || B <= 0 || T <= 0 || bad_pitch(d, ld)) return MI_ERR_ARG;
  const int Th = (T + 1) / 2;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_DT(dys_dtype, TY, MI_LAUNCH((time_recover_bwd_kernel<TY>), dim3(sgrid((long long)B * Th * (ld >> 2))), dim3(256), 0, s,
                                                (const float*)dx, (TY*)dys, B, T, Th, d, ld));
  return mi_check_launch();
}
