// HBM-bound glue kernels of the Conformer block (all vectorised 4-wide, wave64 shuffle reductions):
//   GLU(+pad mask), dropout-scale-cast, (q+u | q+v) bias broadcast, rel-pos score assembly + masked softmax (+dropout)
//   and its backward (including the "un-skew" of the rel_shift), small add kernels.
//
// Replaces on the reference path:
//   nn.functional.glu + masked_fill            conformer_modules.py:324-331
//   nn.Dropout on branch outputs               conformer_modules.py:177,192,207,212  (mask regenerated, never stored)
//   q + pos_bias_u / q + pos_bias_v            multi_head_attention.py:305-307
//   rel_shift + (ac+bd)/sqrt(dk) + masked softmax + dropout     multi_head_attention.py:259-270,343-346,137-140
#include "common.h"
#include "mi355x_asr.h"

#define DISPATCH_DT(dt, T, ...)                                      \
  if ((dt) == MI_DT_F32) { typedef float T; __VA_ARGS__; }           \
  else { typedef bf16_t T; __VA_ARGS__; }

// ------------------------------------------------------------------------------------------------ GLU
// in [M, 2d] -> out [M, d] = a * sigmoid(b) * (t < len[b]);  rows m = b*T + t
// cu != nullptr ("packed rows"): `in` holds only the valid frames, utterance b at rows cu[b] .. cu[b] + len[b] - 1;
// `out` stays the padded [B*T, d] grid the depthwise convolution / BatchNorm run on (their statistics cover padded frames too).
template <typename T>
__global__ __launch_bounds__(256) void glu_fwd_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                      const long long* __restrict__ len, int Tt, long long M, int d,
                                                      const long long* __restrict__ cu) {
  const long long nv = M * (d >> 2);
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    const long long m = i / (d >> 2);
    const int c = (int)(i - m * (d >> 2)) * 4;
    const int b = (int)(m / Tt), t = (int)(m - (long long)b * Tt);
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (!len || t < len[b]) {
      float a[4], g[4];
      const long long mi = cu ? cu[b] + t : m;
      ld4(in + mi * 2 * d + c, a); ld4(in + mi * 2 * d + d + c, g);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = a[j] * sigmoidf_(g[j]);
    }
    st4(out + m * d + c, o);
  }
}
template <typename T>
// (cu: `in` and `din` are packed, `dout` is the padded grid; a frame beyond its utterance has no row in `din`)
__global__ __launch_bounds__(256) void glu_bwd_kernel(const T* __restrict__ in, const T* __restrict__ dout, T* __restrict__ din,
                                                      const long long* __restrict__ len, int Tt, long long M, int d,
                                                      const long long* __restrict__ cu) {
  const long long nv = M * (d >> 2);
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    const long long m = i / (d >> 2);
    const int c = (int)(i - m * (d >> 2)) * 4;
    const int b = (int)(m / Tt), t = (int)(m - (long long)b * Tt);
    float da[4] = {0.f, 0.f, 0.f, 0.f}, dg[4] = {0.f, 0.f, 0.f, 0.f};
    const bool valid = !len || t < len[b];
    if (cu && !valid) continue;
    const long long mi = cu ? cu[b] + t : m;
    if (valid) {
      float a[4], g[4], e[4];
      ld4(in + mi * 2 * d + c, a); ld4(in + mi * 2 * d + d + c, g); ld4(dout + m * d + c, e);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = sigmoidf_(g[j]);
        da[j] = e[j] * s;
        dg[j] = e[j] * a[j] * s * (1.f - s);
      }
    }
    st4(din + mi * 2 * d + c, da); st4(din + mi * 2 * d + d + c, dg);
  }
}

// ------------------------------------------------------------------------------------------------ packed rows <-> padded grid
// PACK  : packed[cu[b] + t, 0:w] = padded[b*T + t, 0:w]                      for t < len[b]
// UNPACK: padded[b*T + t, 0:w] = t < len[b] ? packed[cu[b] + t, 0:w] : 0     for every t < T
// (row pitches in elements; w a multiple of the 16-byte vector: 8 bf16 / 4 f32).  "Packed" = the valid frames of every utterance
// back to back -- what the row-wise chain (LayerNorm, feed-forward, projections, residuals) runs on when the batch is ragged.
template <typename T, bool UNPACK>
__global__ __launch_bounds__(256) void rows_pack_kernel(const T* __restrict__ src, T* __restrict__ dst, long long ld_src,
                                                        long long ld_dst, const long long* __restrict__ len,
                                                        const long long* __restrict__ cu, int Tt, long long M, int w) {
  constexpr int V = VecIO<T>::V;
  const int wv = w / V;
  const long long nv = M * wv;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    const long long m = i / wv;
    const int c = (int)(i - m * wv) * V;
    const int b = (int)(m / Tt), t = (int)(m - (long long)b * Tt);
    const bool valid = t < len[b];
    float v[V];
    if (UNPACK) {
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] = 0.f;
      if (valid) VecIO<T>::load(src + (cu[b] + t) * ld_src + c, v);
      VecIO<T>::store(dst + m * ld_dst + c, v);
    } else if (valid) {
      VecIO<T>::load(src + m * ld_src + c, v);
      VecIO<T>::store(dst + (cu[b] + t) * ld_dst + c, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------ dropout-scale-cast
// out[m,n] = alpha * dropmask(m*N+n) * in[m,n]     (in f32 [M,N] dense, out T dense)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void drop_scale_cast_kernel(const TI* __restrict__ in, TO* __restrict__ out, long long n8,
                                                              float alpha, DropCfg drop) {
  drop_resolve(drop);
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    float a[4], b[4], m[8];
    ld4(in + i * 8, a); ld4(in + i * 8 + 4, b);
    drop_mask8(drop, (uint32_t)(i * 8), m);
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] *= alpha * m[j]; b[j] *= alpha * m[j + 4]; }
    st4(out + i * 8, a); st4(out + i * 8 + 4, b);
  }
}

// ------------------------------------------------------------------------------------------------ q + u, q + v
// qkv [M, ldq] (q in the first d columns) ; u, v f32 [d] ; qu, qv [M, d]
template <typename T>
__global__ __launch_bounds__(256) void qbias_kernel(const T* __restrict__ qkv, long long ldq, const float* __restrict__ u,
                                                    const float* __restrict__ v, T* __restrict__ qu, T* __restrict__ qv,
                                                    long long M, int d) {
  const long long nv = M * (d >> 2);
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    const long long m = i / (d >> 2);
    const int c = (int)(i - m * (d >> 2)) * 4;
    float q[4], a[4], b[4], o1[4], o2[4];
    ld4(qkv + m * ldq + c, q); ld4(u + c, a); ld4(v + c, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) { o1[j] = q[j] + a[j]; o2[j] = q[j] + b[j]; }
    st4(qu + m * d + c, o1); st4(qv + m * d + c, o2);
  }
}
// out[m, 0:d] (pitch ldo) = a[m,:] + b[m,:]
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void add2_kernel(const TI* __restrict__ a, const TI* __restrict__ b, TO* __restrict__ out,
                                                   long long ldo, long long M, int d) {
  const long long nv = M * (d >> 2);
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    const long long m = i / (d >> 2);
    const int c = (int)(i - m * (d >> 2)) * 4;
    float x[4], y[4];
    ld4(a + m * d + c, x); ld4(b + m * d + c, y);
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] += y[j];
    st4(out + m * ldo + c, x);
  }
}

// ------------------------------------------------------------------------------------------------ rel-pos softmax
// One wave per score row (h, b, i).  ac [H,B,T,Tp] f32, bdf [H,B,T,Pp] f32 (bd_full before rel_shift),
//   score[j] = (ac[j] + bdf[T-1+j-i]) * scale ; masked (i or j beyond len[b]) -> -10000 ; softmax ; masked -> 0
// s (pre-dropout, kept for backward) and pd = dropout(s) are written with pitch Tp, pad columns zeroed.
#define SM_MAXV 16  // supports T <= 64*16 = 1024 frames after subsampling (40 s of audio)
// Limited attention context (ConformerEncoder._create_masks, conformer_encoder.py:794-823; att_context_size = [left, right]):
// style 1 'regular': -left <= j - i <= right (each side only if >= 0); style 2 'chunked_limited': the keys of the query's own chunk
// (chunk = right + 1 frames) and of the left / chunk chunks before it -- with right == -1 the left-limited regular mask.  A key
// outside the window is treated exactly like a padded one: score -10000 into the softmax, probability 0 out of it
// (multi_head_attention.py:137-146).
__device__ __forceinline__ bool ctx_allows(int style, int left, int right, int i, int j) {
  if (style == 1) return (left < 0 || j - i >= -left) && (right < 0 || j - i <= right);
  if (style == 2) {
    if (right < 0) return left < 0 || j - i >= -left;
    const int chunk = right + 1, dc = i / chunk - j / chunk;
    return dc >= 0 && dc <= (left >= 0 ? left / chunk : 10000);
  }
  return true;
}
template <typename TO>
__global__ __launch_bounds__(256) void relpos_softmax_fwd_kernel(const float* __restrict__ ac, const float* __restrict__ bdf,
                                                                 TO* __restrict__ s_out, TO* __restrict__ pd_out,
                                                                 const long long* __restrict__ len, int H, int B, int T, int Tp,
                                                                 int Pp, float scale, DropCfg drop, int ctx_style, int ctx_left,
                                                                 int ctx_right) {
  drop_resolve(drop);
  const int lane = threadIdx.x & 63;
  const long long row = blockIdx.x * 4LL + (threadIdx.x >> 6);
  if (row >= (long long)H * B * T) return;
  const int i = (int)(row % T);
  const int b = (int)((row / T) % B);
  const int L = (int)min((long long)T, len[b]);
  const float* acr = ac + row * Tp;
  const float* bdr = bdf + row * Pp + (T - 1 - i);
  const bool row_valid = i < L;
  float v[SM_MAXV];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < SM_MAXV; ++k) {
    const int j = lane + k * 64;
    float sc = -INFINITY;
    if (j < T) sc = (row_valid && j < L && ctx_allows(ctx_style, ctx_left, ctx_right, i, j)) ? (acr[j] + bdr[j]) * scale : -10000.f;
    v[k] = sc;
    mx = fmaxf(mx, sc);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < SM_MAXV; ++k) {
    const int j = lane + k * 64;
    float e = (j < T) ? __expf(v[k] - mx) : 0.f;
    v[k] = e;
    sum += e;
  }
  const float inv = 1.f / wave_sum(sum);
#pragma unroll
  for (int k = 0; k < SM_MAXV; ++k) {
    const int j = lane + k * 64;
    if (j < Tp) {
      float p = (j < T && row_valid && j < L && ctx_allows(ctx_style, ctx_left, ctx_right, i, j)) ? v[k] * inv : 0.f;
      st(s_out + row * Tp + j, p);
      if (pd_out) st(pd_out + row * Tp + j, p * drop_mask(drop, (uint32_t)(row * Tp + j)));
    }
  }
}

// backward: dpd [H,B,T,Tp] (d loss / d dropout(s)), s -> dscore [H,B,T,Tp] (= d ac) and dbdf [H,B,T,Pp] (un-skewed d bd)
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void relpos_softmax_bwd_kernel(const TD* __restrict__ dpd, const TS* __restrict__ s_in,
                                                                 TS* __restrict__ dscore, TS* __restrict__ dbdf, int H, int B,
                                                                 int T, int Tp, int Pp, float scale, DropCfg drop) {
  drop_resolve(drop);
  const int lane = threadIdx.x & 63;
  const long long row = blockIdx.x * 4LL + (threadIdx.x >> 6);
  if (row >= (long long)H * B * T) return;
  const int i = (int)(row % T);
  float sv[SM_MAXV], dp[SM_MAXV];
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < SM_MAXV; ++k) {
    const int j = lane + k * 64;
    float s = 0.f, g = 0.f;
    if (j < T) {
      s = ld(s_in + row * Tp + j);
      g = ld(dpd + row * Tp + j) * drop_mask(drop, (uint32_t)(row * Tp + j));
    }
    sv[k] = s; dp[k] = g;
    dot += s * g;
  }
  dot = wave_sum(dot);
  // zero the un-skewed row first (only T of its Pp entries are non-zero: c in [T-1-i, 2T-2-i])
  TS* dbr = dbdf + row * Pp;
  const int c_lo = T - 1 - i;
  for (int c = lane; c < Pp; c += 64)
    if (c < c_lo || c >= c_lo + T) st(dbr + c, 0.f);
#pragma unroll
  for (int k = 0; k < SM_MAXV; ++k) {
    const int j = lane + k * 64;
    if (j < Tp) {
      const float ds = (j < T) ? sv[k] * (dp[k] - dot) * scale : 0.f;
      st(dscore + row * Tp + j, ds);
      if (j < T) st(dbr + c_lo + j, ds);
    }
  }
}

// x[r, :] *= vec[r]   (per-utterance upstream gradient of the CTC loss)
__global__ __launch_bounds__(256) void row_scale_kernel(float* __restrict__ x, const float* __restrict__ vec, long long rows,
                                                        long long cols) {
  const long long n = rows * cols;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) x[i] *= vec[i / cols];
}

// =================================================================================================
static inline int grid_for(long long n) { long long g = (n + 255) / 256; return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }

extern "C" int mi355x_glu_fwd(const void* in, void* out, int dt, const void* len, int T, long long M, int d,
                              const void* row_offsets, void* stream) {
  mi_clear_errors();
  if (!in || !out || M <= 0 || d <= 0 || (d & 3) || T <= 0 || (row_offsets && !len)) return MI_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_DT(dt, TT, MI_LAUNCH((glu_fwd_kernel<TT>), dim3(grid_for(M * (d >> 2))), dim3(256), 0, s, (const TT*)in,
                                         (TT*)out, (const long long*)len, T, M, d, (const long long*)row_offsets));
  return mi_check_launch();
}
extern "C" int mi355x_glu_bwd(const void* in, const void* dout, void* din, int dt, const void* len, int T, long long M, int d,
                              const void* row_offsets, void* stream) {
  mi_clear_errors();
  if (!in || !dout || !din || M <= 0 || d <= 0 || (d & 3) || T <= 0 || (row_offsets && !len)) return MI_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_DT(dt, TT, MI_LAUNCH((glu_bwd_kernel<TT>), dim3(grid_for(M * (d >> 2))), dim3(256), 0, s, (const TT*)in,
                                         (const TT*)dout, (TT*)din, (const long long*)len, T, M, d, (const long long*)row_offsets));
  return mi_check_launch();
}
// direction 0: padded -> packed, 1: packed -> padded (zero rows beyond the utterances); M = B * T rows of the PADDED grid
extern "C" int mi355x_rows_pack(const void* src, void* dst, int dt, long long ld_src, long long ld_dst, const void* len,
                                const void* row_offsets, int T, long long M, int width, int direction, void* stream) {
  mi_clear_errors();
  if (!src || !dst || !len || !row_offsets || M <= 0 || T <= 0 || width <= 0 || (direction != 0 && direction != 1)) return MI_ERR_ARG;
  const int V = dt == MI_DT_F32 ? 4 : 8;
  if ((width % V) || (ld_src % V) || (ld_dst % V) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return MI_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long long nv = M * (width / V);
  if (direction == 0) {
    DISPATCH_DT(dt, TT, MI_LAUNCH((rows_pack_kernel<TT, false>), dim3(grid_for(nv)), dim3(256), 0, s, (const TT*)src, (TT*)dst, ld_src,
                                  ld_dst, (const long long*)len, (const long long*)row_offsets, T, M, width));
  } else {
    DISPATCH_DT(dt, TT, MI_LAUNCH((rows_pack_kernel<TT, true>), dim3(grid_for(nv)), dim3(256), 0, s, (const TT*)src, (TT*)dst, ld_src,
                                  ld_dst, (const long long*)len, (const long long*)row_offsets, T, M, width));
  }
  return mi_check_launch();
}
extern "C" int mi355x_drop_scale_cast(const void* in, int in_dt, void* out, int out_dt, long long n, float alpha,
                                      unsigned drop_key, unsigned drop_threshold, float drop_scale, void* stream) {
  mi_clear_errors();
  if (!in || !out || n <= 0 || (n & 7)) return MI_ERR_ARG;
  DropCfg dc = mi_drop(drop_key, drop_threshold, drop_scale);
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_DT(in_dt, TI, DISPATCH_DT(out_dt, TO,
    MI_LAUNCH((drop_scale_cast_kernel<TI, TO>), dim3(grid_for(n >> 3)), dim3(256), 0, s, (const TI*)in, (TO*)out,
                       n >> 3, alpha, dc)));
  return mi_check_launch();
}
extern "C" int mi355x_qbias(const void* qkv, long long ldq, const void* u, const void* v, void* qu, void* qv, int dt,
                            long long M, int d, void* stream) {
  mi_clear_errors();
  if (!qkv || !u || !v || !qu || !qv || M <= 0 || (d & 3) || (ldq & 3)) return MI_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_DT(dt, TT, MI_LAUNCH((qbias_kernel<TT>), dim3(grid_for(M * (d >> 2))), dim3(256), 0, s, (const TT*)qkv, ldq,
                                         (const float*)u, (const float*)v, (TT*)qu, (TT*)qv, M, d));
  return mi_check_launch();
}
// out[m, 0:d] = a + b (row pitch ldo) and the column sums of a and of b in the same pass (dq = dqu + dqv together with the
// pos_bias_u / pos_bias_v gradients, multi_head_attention.py:288-291).  Thread = 8 channels x every 4th row of a 32-row
// block; per-block column sums go to a scratch slab and partials_reduce_kernel adds the slabs (no same-address atomics).
#define A2C_ROWS 32
__global__ __launch_bounds__(256) void add2_colsum_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                          bf16_t* __restrict__ out, long long ldo, long long M, int d,
                                                          float* __restrict__ partial) {
  __shared__ float sred[2][256 * 8];
  const int CP = min(d / 8, 256), RS = 256 / CP;
  const int ck = threadIdx.x % CP, rsub = threadIdx.x / CP;
  const long long r0 = (long long)blockIdx.x * A2C_ROWS, r1 = min(M, r0 + A2C_ROWS);
  for (int c0 = 0; c0 < d; c0 += CP * 8) {
    const int c = c0 + ck * 8;
    float sa[8], sb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sa[j] = 0.f; sb[j] = 0.f; }
    if (rsub < RS && c < d) {
#pragma unroll 4
      for (long long r = r0 + rsub; r < r1; r += RS) {
        float x[8], y[8];
        VecIO<bf16_t>::load(a + r * d + c, x);
        VecIO<bf16_t>::load(b + r * d + c, y);
#pragma unroll
        for (int j = 0; j < 8; ++j) { sa[j] += x[j]; sb[j] += y[j]; x[j] += y[j]; }
        VecIO<bf16_t>::store(out + r * ldo + c, x);
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) { sred[0][threadIdx.x * 8 + j] = sa[j]; sred[1][threadIdx.x * 8 + j] = sb[j]; }
    __syncthreads();
    for (int e = threadIdx.x; e < CP * 8; e += 256) {
      float t1 = 0.f, t2 = 0.f;
      for (int q = 0; q < RS; ++q) { t1 += sred[0][q * CP * 8 + e]; t2 += sred[1][q * CP * 8 + e]; }
      if (c0 + e < d) {
        partial[((long long)blockIdx.x * 2 + 0) * d + c0 + e] = t1;
        partial[((long long)blockIdx.x * 2 + 1) * d + c0 + e] = t2;
      }
    }
  }
}
extern "C" int mi355x_add2_colsum(const void* a, const void* b, void* out, long long ldo, long long M, int d, void* sum_ab,
                                  void* scratch, long long scratch_elems, void* stream) {
  mi_clear_errors();
  if (!a || !b || !out || !sum_ab || !scratch || M <= 0 || d <= 0 || (d & 7) || (ldo & 7)) return MI_ERR_ARG;
  const int nblk = (int)((M + A2C_ROWS - 1) / A2C_ROWS);
  if (scratch_elems < (long long)nblk * 2 * d) return MI_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  MI_LAUNCH(add2_colsum_kernel, dim3(nblk), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, ldo, M,
                     d, (float*)scratch);
  MI_LAUNCH((partials_reduce_kernel<float>), dim3((2 * d + 255) / 256, 32), dim3(256), 0, s, (const float*)scratch, nblk,
                     2 * d, (float*)sum_ab);
  return mi_check_launch();
}
extern "C" int mi355x_add2(const void* a, const void* b, int in_dt, void* out, int out_dt, long long ldo, long long M, int d,
                           void* stream) {
  mi_clear_errors();
  if (!a || !b || !out || M <= 0 || (d & 3) || (ldo & 3)) return MI_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_DT(in_dt, TI, DISPATCH_DT(out_dt, TO,
    MI_LAUNCH((add2_kernel<TI, TO>), dim3(grid_for(M * (d >> 2))), dim3(256), 0, s, (const TI*)a, (const TI*)b,
                       (TO*)out, ldo, M, d)));
  return mi_check_launch();
}
extern "C" int mi355x_relpos_softmax_fwd_ctx(const void* ac, const void* bdf, void* s_out, void* pd_out, int out_dt, const void* len,
                                             int H, int B, int T, int Tp, int Pp, float scale, unsigned drop_key,
                                             unsigned drop_threshold, float drop_scale, int ctx_style, int ctx_left, int ctx_right,
                                             void* stream) {
  mi_clear_errors();
  if (!ac || !bdf || !s_out || !len || T <= 0 || T > 64 * SM_MAXV || Tp < T || Tp > 64 * SM_MAXV || Pp < 2 * T - 1)
    return MI_ERR_ARG;
  if (ctx_style < 0 || ctx_style > 2 || ctx_left < -1 || ctx_right < -1) return MI_ERR_ARG;
  DropCfg dc = mi_drop(drop_key, drop_threshold, drop_scale);
  const long long rows = (long long)H * B * T;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_DT(out_dt, TO, MI_LAUNCH((relpos_softmax_fwd_kernel<TO>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s,
                                             (const float*)ac, (const float*)bdf, (TO*)s_out, (TO*)pd_out,
                                             (const long long*)len, H, B, T, Tp, Pp, scale, dc, ctx_style, ctx_left, ctx_right));
  return mi_check_launch();
}
extern "C" int mi355x_relpos_softmax_fwd(const void* ac, const void* bdf, void* s_out, void* pd_out, int out_dt, const void* len,
                                         int H, int B, int T, int Tp, int Pp, float scale, unsigned drop_key,
                                         unsigned drop_threshold, float drop_scale, void* stream) {
  return mi355x_relpos_softmax_fwd_ctx(ac, bdf, s_out, pd_out, out_dt, len, H, B, T, Tp, Pp, scale, drop_key, drop_threshold,
                                       drop_scale, 0, -1, -1, stream);
}
extern "C" int mi355x_relpos_softmax_bwd(const void* dpd, int dpd_dt, const void* s_in, void* dscore, void* dbdf, int s_dt, int H,
                                         int B, int T, int Tp, int Pp, float scale, unsigned drop_key, unsigned drop_threshold,
                                         float drop_scale, void* stream) {
  mi_clear_errors();
  if (!dpd || !s_in || !dscore || !dbdf || T <= 0 || T > 64 * SM_MAXV || Tp < T || Tp > 64 * SM_MAXV || Pp < 2 * T - 1)
    return MI_ERR_ARG;
  DropCfg dc = mi_drop(drop_key, drop_threshold, drop_scale);
  const long long rows = (long long)H * B * T;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_DT(s_dt, TS, DISPATCH_DT(dpd_dt, TD,
    MI_LAUNCH((relpos_softmax_bwd_kernel<TS, TD>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (const TD*)dpd,
                       (const TS*)s_in, (TS*)dscore, (TS*)dbdf, H, B, T, Tp, Pp, scale, dc)));
  return mi_check_launch();
}
extern "C" int mi355x_row_scale(void* x, const void* vec, long long rows, long long cols, void* stream) {
  mi_clear_errors();
  if (!x || !vec || rows <= 0 || cols <= 0) return MI_ERR_ARG;
  MI_LAUNCH(row_scale_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, (hipStream_t)stream, (float*)x,
                     (const float*)vec, rows, cols);
  return mi_check_launch();
}

// ------------------------------------------------------------------------------------------------ SpecAugment / SpecCutout
// x[b, f0:f1, t0:t1] = value for a device-resident list of rectangles (b, f0, f1, t0, t1) -- the time masks, frequency
// masks and cut-out rectangles of SpectrogramAugmentation (spectr_augment.py:134-215, 245-261) applied in ONE launch
// that touches only the masked elements (the reference materialises a full boolean mask and runs masked_fill twice).
// grid (rects, 64 row-blocks): one workgroup row-block per 'f' group, threads along t.
__global__ __launch_bounds__(256) void fill_rects_kernel(float* __restrict__ x, const int* __restrict__ rects, int n, int B, int F,
                                                         int T, float value) {
  const int r = blockIdx.x;
  if (r >= n) return;
  const int b = rects[5 * r], f0 = max(rects[5 * r + 1], 0), f1 = min(rects[5 * r + 2], F);
  const int t0 = max(rects[5 * r + 3], 0), t1 = min(rects[5 * r + 4], T);
  if (b < 0 || b >= B || f0 >= f1 || t0 >= t1) return;
  for (int f = f0 + blockIdx.y; f < f1; f += gridDim.y) {
    float* row = x + ((long long)b * F + f) * T;
    for (int t = t0 + threadIdx.x; t < t1; t += 256) row[t] = value;
  }
}
extern "C" int mi355x_fill_rects(void* x, const void* rects, int n, int B, int F, int T, float value, void* stream) {
  mi_clear_errors();
  if (!x || B <= 0 || F <= 0 || T <= 0 || n < 0 || (n > 0 && !rects)) return MI_ERR_ARG;
  if (n == 0) return 0;
  MI_LAUNCH(fill_rects_kernel, dim3(n, F < 64 ? F : 64), dim3(256), 0, (hipStream_t)stream, (float*)x, (const int*)rects,
                     n, B, F, T, value);
  return mi_check_launch();
}

// SpecAugment mask parameters on the device (round 5): the reference's vectorised path (spectr_augment.py:155-195) turns four
// uniform draws into mask widths / starts with ~40 tiny tensor ops (multiply, clamp, .long(), subtract, stack, cat).  The draws
// stay four torch.rand calls on the caller's side (same generator stream as the reference); this kernel does the rest in the same
// f32 arithmetic -- width = min(f32(len) * tw, T) for a float time_width, w = trunc(u1 * width), start = trunc(u2 * f32(len - w)),
// frequency masks against F -- and writes the rectangle table mi355x_fill_rects reads: rows (b, f0, f1, t0, t1), time masks
// first (utterance-major), then frequency masks.
__global__ __launch_bounds__(256) void specaug_rects_kernel(const float* __restrict__ u_tw, const float* __restrict__ u_ts,
                                                            const float* __restrict__ u_fw, const float* __restrict__ u_fs,
                                                            const long long* __restrict__ len, int* __restrict__ rects, int B, int nt,
                                                            int nf, int F, int T, float tw, int tw_adaptive, float fw) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * (nt + nf)) return;
  int* r = rects + 5 * e;
  if (e < B * nt) {
    const int b = e / nt;
    const long long L = len[b];
    const float width = tw_adaptive ? fminf((float)L * tw, (float)T) : tw;
    const long long mw = (long long)(u_tw[e] * width);
    const long long ms = (long long)(u_ts[e] * (float)(L - mw));
    r[0] = b; r[1] = 0; r[2] = F; r[3] = (int)ms; r[4] = (int)(ms + mw);
  } else {
    const int e2 = e - B * nt;
    const int b = e2 / nf;
    const long long mw = (long long)(u_fw[e2] * fw);
    const long long ms = (long long)(u_fs[e2] * (float)((long long)F - mw));
    r[0] = b; r[1] = (int)ms; r[2] = (int)(ms + mw); r[3] = 0; r[4] = T;
  }
}
extern "C" int mi355x_specaug_rects(const void* u_time_width, const void* u_time_start, const void* u_freq_width,
                                    const void* u_freq_start, const void* len, void* rects, int B, int time_masks, int freq_masks,
                                    int F, int T, float time_width, int time_width_is_fraction, int freq_width, void* stream) {
  mi_clear_errors();
  if (!len || !rects || B <= 0 || F <= 0 || T <= 0 || time_masks < 0 || freq_masks < 0 || time_masks + freq_masks == 0) return MI_ERR_ARG;
  if ((time_masks > 0 && (!u_time_width || !u_time_start)) || (freq_masks > 0 && (!u_freq_width || !u_freq_start))) return MI_ERR_ARG;
  const int n = B * (time_masks + freq_masks);
  MI_LAUNCH(specaug_rects_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)u_time_width,
            (const float*)u_time_start, (const float*)u_freq_width, (const float*)u_freq_start, (const long long*)len, (int*)rects, B,
            time_masks, freq_masks, F, T, time_width, time_width_is_fraction, (float)freq_width);
  return mi_check_launch();
}
