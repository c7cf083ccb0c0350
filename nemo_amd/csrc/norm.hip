// LayerNorm / log-softmax / column-reduction kernels (HBM-bound; wave64 shuffle reductions, 16-B vector access).
//
// Replaces on the reference path: torch.nn.LayerNorm x5 per ConformerLayer
// (nemo/collections/asr/parts/submodules/conformer_modules.py:98,102,115,152,157 used at :174-215),
// log_softmax of the decoder (nemo/collections/asr/modules/conv_asr.py:468) and the bias-gradient column sums
// autograd produces for every Linear / Conv1d on the path.
#include "common.h"
#include "mi355x_asr.h"

#define DISPATCH_DT(dt, T, ...)                                      \
  if ((dt) == MI_DT_F32) { typedef float T; __VA_ARGS__; }           \
  else { typedef bf16_t T; __VA_ARGS__; }

// ------------------------------------------------------------------------------------------------
// LayerNorm forward: one wave per row, 4 rows per 256-thread block.  Two-pass (mean, then centred variance) on
// register-resident data when d <= 64*4*VPL, else re-read.  Statistics saved for backward.
// ------------------------------------------------------------------------------------------------
// 8 consecutive elements <-> f32 registers (bf16: one 16-byte access, f32: two)
__device__ __forceinline__ void ld8g(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void ld8g(const bf16_t* p, float (&v)[8]) { VecIO<bf16_t>::load(p, v); }
__device__ __forceinline__ void st8g(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void st8g(bf16_t* p, const float (&v)[8]) { VecIO<bf16_t>::store(p, v); }

template <typename TX, typename TY>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TX* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, TY* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int M, int d,
                                                     float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const TX* xr = x + (long long)row * d;
  TY* yr = y + (long long)row * d;
  const int nv = d >> 2;  // d % 4 == 0 guaranteed by the launcher
  float s = 0.f;
  for (int i = lane; i < nv; i += 64) {
    float v[4]; ld4(xr + i * 4, v);
    s += (v[0] + v[1]) + (v[2] + v[3]);
  }
  const float mu = wave_sum(s) / (float)d;
  float q = 0.f;
  for (int i = lane; i < nv; i += 64) {
    float v[4]; ld4(xr + i * 4, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) { float c = v[j] - mu; q += c * c; }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)d + eps);
  for (int i = lane; i < nv; i += 64) {
    float v[4], g[4], b[4], o[4];
    ld4(xr + i * 4, v); ld4(gamma + i * 4, g); ld4(beta + i * 4, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (v[j] - mu) * rs * g[j] + b[j];
    st4(yr + i * 4, o);
  }
  if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
}

// LayerNorm backward wrt input: dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat));  dres (+)= dx
template <typename TX, typename TDY>
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(const TDY* __restrict__ dy, const TX* __restrict__ x,
                                                        const float* __restrict__ gamma, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, float* __restrict__ dres,
                                                        int accumulate, int M, int d) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const TX* xr = x + (long long)row * d;
  const TDY* dyr = dy + (long long)row * d;
  float* dr = dres + (long long)row * d;
  const float mu = mean[row], rs = rstd[row];
  const int nv = d >> 2;
  float s1 = 0.f, s2 = 0.f;
  for (int i = lane; i < nv; i += 64) {
    float v[4], g[4], e[4];
    ld4(xr + i * 4, v); ld4(gamma + i * 4, g); ld4(dyr + i * 4, e);
#pragma unroll
    for (int j = 0; j < 4; ++j) { float gd = g[j] * e[j]; s1 += gd; s2 += gd * (v[j] - mu) * rs; }
  }
  s1 = wave_sum(s1) / (float)d;
  s2 = wave_sum(s2) / (float)d;
  for (int i = lane; i < nv; i += 64) {
    float v[4], g[4], e[4], o[4];
    ld4(xr + i * 4, v); ld4(gamma + i * 4, g); ld4(dyr + i * 4, e);
    if (accumulate) ld4(dr + i * 4, o); else { o[0] = o[1] = o[2] = o[3] = 0.f; }
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] += rs * (g[j] * e[j] - s1 - (v[j] - mu) * rs * s2);
    st4(dr + i * 4, o);
  }
}

// Fused LayerNorm backward: dx (as above) AND the parameter gradients in one pass over dy / x.  A block walks LNB_ROWS rows
// (wave w takes rows w, w+4, ...); every lane owns the same NV 4-column groups for all its rows, so dgamma / dbeta
// partials live in registers, are combined across the 4 waves through LDS and leave as one atomic per column per block.
#define LNB_ROWS 64
#define LNB_WAVES 16  // 1024-thread workgroups: 16 rows in flight per CU, ONE atomic per column per 64 rows (the
                      // parameter-gradient atomics all hit the same 2*d addresses: fewer, fatter workgroups win)
template <typename TX, typename TDY, int NV>
__global__ __launch_bounds__(64 * LNB_WAVES) void ln_bwd_fused_kernel(const TDY* __restrict__ dy, const TX* __restrict__ x,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, float* __restrict__ dres,
                                                           int accumulate, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int M, int d) {
  __shared__ float red[LNB_WAVES][2][NV * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = d >> 2;
  float g[NV][4], ag[NV][4], ab[NV][4];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = lane + 64 * k;
    if (i < nv) ld4(gamma + i * 4, g[k]);
#pragma unroll
    for (int j = 0; j < 4; ++j) { ag[k][j] = 0.f; ab[k][j] = 0.f; }
  }
  const int r0 = blockIdx.x * LNB_ROWS, r1 = min(M, r0 + LNB_ROWS);
  for (int row = r0 + wave; row < r1; row += LNB_WAVES) {
    const TX* xr = x + (long long)row * d;
    const TDY* dyr = dy + (long long)row * d;
    float* dr = dres + (long long)row * d;
    const float mu = mean[row], rs = rstd[row];
    float xh[NV][4], e[NV][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = lane + 64 * k;
      if (i < nv) {
        float v[4];
        ld4(xr + i * 4, v); ld4(dyr + i * 4, e[k]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xh[k][j] = (v[j] - mu) * rs;
          const float gd = g[k][j] * e[k][j];
          s1 += gd; s2 += gd * xh[k][j];
          ag[k][j] += e[k][j] * xh[k][j]; ab[k][j] += e[k][j];
        }
      }
    }
    s1 = wave_sum(s1) / (float)d;
    s2 = wave_sum(s2) / (float)d;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = lane + 64 * k;
      if (i < nv) {
        float o[4];
        if (accumulate) ld4(dr + i * 4, o); else { o[0] = o[1] = o[2] = o[3] = 0.f; }
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] += rs * (g[k][j] * e[k][j] - s1 - xh[k][j] * s2);
        st4(dr + i * 4, o);
      }
    }
  }
  if (!dgamma) return;
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      red[wave][0][(lane + 64 * k) * 4 + j] = ag[k][j];
      red[wave][1][(lane + 64 * k) * 4 + j] = ab[k][j];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += 64 * LNB_WAVES) {
    float sg_ = 0.f, sb_ = 0.f;
#pragma unroll
    for (int w = 0; w < LNB_WAVES; ++w) { sg_ += red[w][0][c]; sb_ += red[w][1][c]; }
    atomicAdd(dgamma + c, sg_);
    atomicAdd(dbeta + c, sb_);
  }
}

// dgamma[n] += sum_m dy*xhat ; dbeta[n] += sum_m dy.   Block = 64 columns x 4 row-lanes, ROWS_PER_BLOCK rows.
#define CR_ROWS 256
template <typename TX, typename TDY>
__global__ __launch_bounds__(256) void ln_bwd_param_kernel(const TDY* __restrict__ dy, const TX* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int M,
                                                           int d) {
  __shared__ float sg[4][64], sb[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * CR_ROWS, r1 = min(M, r0 + CR_ROWS);
  float ag = 0.f, ab = 0.f;
  if (c < d)
    for (int r = r0 + rl; r < r1; r += 4) {
      float e = ld(dy + (long long)r * d + c);
      float xh = (ld(x + (long long)r * d + c) - mean[r]) * rstd[r];
      ag += e * xh; ab += e;
    }
  sg[rl][threadIdx.x & 63] = ag; sb[rl][threadIdx.x & 63] = ab;
  __syncthreads();
  if (rl == 0 && c < d) {
    const int t = threadIdx.x;
    atomicAdd(dgamma + c, (sg[0][t] + sg[1][t]) + (sg[2][t] + sg[3][t]));
    atomicAdd(dbeta + c, (sb[0][t] + sb[1][t]) + (sb[2][t] + sb[3][t]));
  }
}

// out[n] += alpha * sum_m x[m, n]   (bias gradients, pos_bias_u/v gradients)
template <typename TX>
__global__ __launch_bounds__(256) void colsum_kernel(const TX* __restrict__ x, long long ldx_, float* __restrict__ out,
                                                     int M, int N, float alpha) {
  __shared__ float sa[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * CR_ROWS, r1 = min(M, r0 + CR_ROWS);
  float a = 0.f;
  if (c < N)
    for (int r = r0 + rl; r < r1; r += 4) a += ld(x + (long long)r * ldx_ + c);
  sa[rl][threadIdx.x & 63] = a;
  __syncthreads();
  if (rl == 0 && c < N) {
    const int t = threadIdx.x;
    atomicAdd(out + c, alpha * ((sa[0][t] + sa[1][t]) + (sa[2][t] + sa[3][t])));
  }
}
// bf16, 8 columns (16 B) per lane: block = 512 columns x 4 row lanes, 64 rows per block.  With N < 512 (the 256 channels of the
// sub-sampling stack) a wave covers 512 / N rows at once instead of leaving lanes idle.
#define CSV_ROWS 64
__global__ __launch_bounds__(256) void colsum_bf16x8_kernel(const bf16_t* __restrict__ x, long long ldx_, float* __restrict__ out,
                                                            int M, int N, float alpha) {
  __shared__ float sa[4][512];
  const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int nch = (N + 7) >> 3;                                        // 16-byte chunks per row
  const int CL = (nch >= 64 || (64 % nch)) ? 64 : nch, NR = 64 / CL;   // lanes per row, rows per wave pass
  const int lc = lane % CL, sub = lane / CL;
  const int c = blockIdx.x * 512 + lc * 8;
  const int r0 = blockIdx.y * CSV_ROWS, r1 = min(M, r0 + CSV_ROWS);
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < N) {
#pragma unroll 4
    for (int r = r0 + rl * NR + sub; r < r1; r += 4 * NR) {
      const u32x4 t = *reinterpret_cast<const u32x4*>(x + (long long)r * ldx_ + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[2 * j] += __uint_as_float(t[j] << 16); a[2 * j + 1] += __uint_as_float(t[j] & 0xffff0000u); }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) sa[rl][lane * 8 + j] = a[j];
  __syncthreads();
  for (int e = threadIdx.x; e < CL * 8; e += 256) {
    const int cc = blockIdx.x * 512 + e;
    if (cc >= N) continue;
    float v = 0.f;
    for (int s2 = 0; s2 < NR; ++s2) v += (sa[0][s2 * CL * 8 + e] + sa[1][s2 * CL * 8 + e]) + (sa[2][s2 * CL * 8 + e] + sa[3][s2 * CL * 8 + e]);
    atomicAdd(out + cc, alpha * v);
  }
}

// ------------------------------------------------------------------------------------------------
// log-softmax over the class axis (C = V+1 = 129): one wave per row; logits f32 with pitch ld
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void log_softmax_fwd_kernel(const float* __restrict__ x, long long ldx_, float* __restrict__ y,
                                                              long long ldy, int M, int C) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + row * ldx_;
  float mx = -INFINITY;
  for (int i = lane; i < C; i += 64) mx = fmaxf(mx, xr[i]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int i = lane; i < C; i += 64) s += expf(xr[i] - mx);
  const float lse = mx + logf(wave_sum(s));
  for (int i = lane; i < C; i += 64) y[row * ldy + i] = xr[i] - lse;
}
// dx = dy - exp(y) * sum(dy); output in `TO` with pitch ldo (pad columns [C, ldo) are zero-filled so the buffer can
// be used directly as a K-padded GEMM operand)
template <typename TO>
__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                              long long ldy, TO* __restrict__ dx, long long ldo, int M, int C,
                                                              float scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float s = 0.f;
  for (int i = lane; i < C; i += 64) s += dy[row * ldy + i];
  s = wave_sum(s);
  for (int i = lane; i < (int)ldo; i += 64) {
    float v = 0.f;
    if (i < C) v = scale * (dy[row * ldy + i] - expf(y[row * ldy + i]) * s);
    st(dx + row * ldo + i, v);
  }
}

// =================================================================================================
// Register-resident variant for d % 512 == 0, d <= 2048: one wave per row, lane = 8 consecutive elements per 512-wide
// chunk (32-byte loads, 16-byte bf16 / 32-byte f32 stores), the row is read from memory exactly once.
template <typename TX, typename TY, int NCH>
__global__ __launch_bounds__(256) void ln_fwd_reg_kernel(const TX* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, TY* __restrict__ y,
                                                         float* __restrict__ mean, float* __restrict__ rstd, int M, int d,
                                                         float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const TX* xr = x + (long long)row * d;
  TY* yr = y + (long long)row * d;
  float v[NCH][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    ld8g(xr + c * 512 + lane * 8, v[c]);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[c][j];
  }
  const float mu = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float t = v[c][j] - mu; q += t * t; }
  const float rs = rsqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    float g[8], b[8], o[8];
    ld8g(gamma + c * 512 + lane * 8, g); ld8g(beta + c * 512 + lane * 8, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (v[c][j] - mu) * rs * g[j] + b[j];
    st8g(yr + c * 512 + lane * 8, o);
  }
  if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
}

// Two LayerNorms in a row over register-resident rows (d = 512): y1 = LN1(x) in f32 (a layer's norm_out: the next layer's
// residual stream) and y2 = LN2(y1) in the compute dtype (the next layer's norm_feed_forward1: its first GEMM operand) -- the
// block boundary of conformer_modules.py:205-231 without re-reading y1.  Both (mean, rstd) pairs are kept for backward.
template <typename TY2>
__global__ __launch_bounds__(256) void ln2_fwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ gamma1,
                                                          const float* __restrict__ beta1, float* __restrict__ y1,
                                                          float* __restrict__ mean1, float* __restrict__ rstd1,
                                                          const float* __restrict__ gamma2, const float* __restrict__ beta2,
                                                          TY2* __restrict__ y2, float* __restrict__ mean2,
                                                          float* __restrict__ rstd2, int M, float eps) {
  constexpr int d = 512;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float v[8], g[8], b[8], o[8];
  ld8g(x + (long long)row * d + lane * 8, v);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += v[j];
  const float mu = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) { const float t = v[j] - mu; q += t * t; }
  const float rs = rsqrtf(wave_sum(q) / (float)d + eps);
  ld8g(gamma1 + lane * 8, g); ld8g(beta1 + lane * 8, b);
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (v[j] - mu) * rs * g[j] + b[j];
  st8g(y1 + (long long)row * d + lane * 8, o);
  s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += o[j];
  const float mu2 = wave_sum(s) / (float)d;
  q = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) { const float t = o[j] - mu2; q += t * t; }
  const float rs2 = rsqrtf(wave_sum(q) / (float)d + eps);
  ld8g(gamma2 + lane * 8, g); ld8g(beta2 + lane * 8, b);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (o[j] - mu2) * rs2 * g[j] + b[j];
  st8g(y2 + (long long)row * d + lane * 8, v);
  if (lane == 0) { mean1[row] = mu; rstd1[row] = rs; mean2[row] = mu2; rstd2[row] = rs2; }
}
extern "C" int mi355x_layernorm2_fwd(const void* x, const void* gamma1, const void* beta1, void* y1, void* mean1, void* rstd1,
                                     const void* gamma2, const void* beta2, void* y2, int y2_dt, void* mean2, void* rstd2, int M,
                                     int d, float eps, void* stream) {
  mi_clear_errors();
  if (!x || !gamma1 || !beta1 || !y1 || !mean1 || !rstd1 || !gamma2 || !beta2 || !y2 || !mean2 || !rstd2 || M <= 0) return MI_ERR_ARG;
  if (d != 512) return MI_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)y1 | (uintptr_t)y2 | (uintptr_t)gamma1 | (uintptr_t)beta1 | (uintptr_t)gamma2 | (uintptr_t)beta2) & 31)
    return MI_ERR_ARG;
  DISPATCH_DT(y2_dt, TY2, MI_LAUNCH((ln2_fwd_reg_kernel<TY2>), dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                                    (const float*)gamma1, (const float*)beta1, (float*)y1, (float*)mean1, (float*)rstd1,
                                    (const float*)gamma2, (const float*)beta2, (TY2*)y2, (float*)mean2, (float*)rstd2, M, eps));
  return mi_check_launch();
}

extern "C" int mi355x_layernorm_fwd(const void* x, int x_dt, const void* gamma, const void* beta, void* y, int y_dt,
                                    void* mean, void* rstd, int M, int d, float eps, void* stream) {
  mi_clear_errors();
  if (!x || !gamma || !beta || !y || M <= 0 || d <= 0 || (d & 3)) return MI_ERR_ARG;
  dim3 grid((M + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  const bool aligned = !((uintptr_t)x & 31) && !((uintptr_t)y & 31) && !((uintptr_t)gamma & 31) && !((uintptr_t)beta & 31);
#define LN_REG(NCH) DISPATCH_DT(x_dt, TX, DISPATCH_DT(y_dt, TY, MI_LAUNCH((ln_fwd_reg_kernel<TX, TY, NCH>), grid, \
    block, 0, s, (const TX*)x, (const float*)gamma, (const float*)beta, (TY*)y, (float*)mean, (float*)rstd, M, d, eps)))
  if (aligned && d == 512) { LN_REG(1); }
  else if (aligned && d == 1024) { LN_REG(2); }
  else if (aligned && d == 2048) { LN_REG(4); }
  else
  DISPATCH_DT(x_dt, TX, DISPATCH_DT(y_dt, TY,
    MI_LAUNCH((ln_fwd_kernel<TX, TY>), grid, block, 0, s, (const TX*)x, (const float*)gamma, (const float*)beta,
                       (TY*)y, (float*)mean, (float*)rstd, M, d, eps)));
#undef LN_REG
  return mi_check_launch();
}

// 8-wide variant for d = 512 / 1024: lane = 8 consecutive elements per 512-wide chunk (16-byte bf16 / 32-byte f32
// accesses).  It can also emit, in the same pass, the NEXT consumer's operand: cast_out = bf16(scale * dropmask * dres_new)
// -- the residual-branch gradient that the following sub-block's output GEMMs read (conformer_modules.py:205-231 backward;
// saves one read of the fp32 gradient and one launch per sub-block).  Dropout groups = 8 consecutive elements = one lane chunk.
template <typename TX, typename TDY, int NCH>
__global__ __launch_bounds__(64 * LNB_WAVES) void ln_bwd_fused8_kernel(const TDY* __restrict__ dy, const TX* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dres,
                                                            int accumulate, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int M, int d,
                                                            bf16_t* __restrict__ cast_out, float cast_scale, DropCfg cast_drop) {
  drop_resolve(cast_drop);
  __shared__ float red[LNB_WAVES][2][NCH * 512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float g[NCH][8], ag[NCH][8], ab[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    ld8g(gamma + c * 512 + lane * 8, g[c]);
#pragma unroll
    for (int j = 0; j < 8; ++j) { ag[c][j] = 0.f; ab[c][j] = 0.f; }
  }
  const int r0 = blockIdx.x * LNB_ROWS, r1 = min(M, r0 + LNB_ROWS);
  for (int row = r0 + wave; row < r1; row += LNB_WAVES) {
    const TX* xr = x + (long long)row * d;
    const TDY* dyr = dy + (long long)row * d;
    float* dr = dres + (long long)row * d;
    const float mu = mean[row], rs = rstd[row];
    float xh[NCH][8], e[NCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float v[8];
      ld8g(xr + c * 512 + lane * 8, v); ld8g(dyr + c * 512 + lane * 8, e[c]);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xh[c][j] = (v[j] - mu) * rs;
        const float gd = g[c][j] * e[c][j];
        s1 += gd; s2 += gd * xh[c][j];
        ag[c][j] += e[c][j] * xh[c][j]; ab[c][j] += e[c][j];
      }
    }
    s1 = wave_sum(s1) / (float)d;
    s2 = wave_sum(s2) / (float)d;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float o[8];
      if (accumulate) ld8g(dr + c * 512 + lane * 8, o);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += rs * (g[c][j] * e[c][j] - s1 - xh[c][j] * s2);
      st8g(dr + c * 512 + lane * 8, o);
      if (cast_out) {
        float m[8];
        drop_mask8(cast_drop, (uint32_t)row * (uint32_t)d + (uint32_t)(c * 512 + lane * 8), m);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] *= cast_scale * m[j];
        st8g(cast_out + (long long)row * d + c * 512 + lane * 8, o);
      }
    }
  }
  if (!dgamma) return;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[wave][0][c * 512 + lane * 8 + j] = ag[c][j];
      red[wave][1][c * 512 + lane * 8 + j] = ab[c][j];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += 64 * LNB_WAVES) {
    float sg_ = 0.f, sb_ = 0.f;
#pragma unroll
    for (int w = 0; w < LNB_WAVES; ++w) { sg_ += red[w][0][c]; sb_ += red[w][1][c]; }
    atomicAdd(dgamma + c, sg_);
    atomicAdd(dbeta + c, sb_);
  }
}

// Two LayerNorm backwards in a row at a layer boundary (d = 512), the mirror image of ln2_fwd_reg_kernel: layer i+1's
// norm_feed_forward1 (input xo = layer i's output, upstream dy1) and layer i's norm_out (input r4).
//   g   = dres_in + dLN1(dy1; xo)            the complete gradient w.r.t. xo -- stays in registers, never written
//   out = dLN2(g; r4) -> dres_out (f32) and, optionally, cast_out = bf16(cast_scale * dropmask * out) for the next GEMMs
// Both LayerNorms' parameter gradients leave through the same 64-KiB LDS buffer, one after the other.  Saves the write and the
// re-read of g (64 MB per boundary) and a launch.
template <typename TDY>
__global__ __launch_bounds__(64 * LNB_WAVES) void ln2_bwd_fused8_kernel(
    const TDY* __restrict__ dy1, const float* __restrict__ x1, const float* __restrict__ gamma1, const float* __restrict__ mean1,
    const float* __restrict__ rstd1, float* __restrict__ dgamma1, float* __restrict__ dbeta1, const float* __restrict__ dres_in,
    const float* __restrict__ x2, const float* __restrict__ gamma2, const float* __restrict__ mean2,
    const float* __restrict__ rstd2, float* __restrict__ dgamma2, float* __restrict__ dbeta2, float* __restrict__ dres_out, int M,
    bf16_t* __restrict__ cast_out, float cast_scale, DropCfg cast_drop) {
  drop_resolve(cast_drop);
  constexpr int d = 512;
  __shared__ float red[LNB_WAVES][2][d];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float g1[8], g2[8], ag1[8], ab1[8], ag2[8], ab2[8];
  ld8g(gamma1 + lane * 8, g1); ld8g(gamma2 + lane * 8, g2);
#pragma unroll
  for (int j = 0; j < 8; ++j) { ag1[j] = 0.f; ab1[j] = 0.f; ag2[j] = 0.f; ab2[j] = 0.f; }
  const int r0 = blockIdx.x * LNB_ROWS, r1 = min(M, r0 + LNB_ROWS);
  for (int row = r0 + wave; row < r1; row += LNB_WAVES) {
    const long long off = (long long)row * d + lane * 8;
    float v[8], e[8], xh[8], g[8];
    ld8g(x1 + off, v); ld8g(dy1 + off, e); ld8g(dres_in + off, g);
    const float mu1 = mean1[row], rs1 = rstd1[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      xh[j] = (v[j] - mu1) * rs1;
      const float gd = g1[j] * e[j];
      s1 += gd; s2 += gd * xh[j];
      ag1[j] += e[j] * xh[j]; ab1[j] += e[j];
    }
    s1 = wave_sum(s1) / (float)d;
    s2 = wave_sum(s2) / (float)d;
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] += rs1 * (g1[j] * e[j] - s1 - xh[j] * s2);
    ld8g(x2 + off, v);
    const float mu2 = mean2[row], rs2 = rstd2[row];
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      xh[j] = (v[j] - mu2) * rs2;
      const float gd = g2[j] * g[j];
      t1 += gd; t2 += gd * xh[j];
      ag2[j] += g[j] * xh[j]; ab2[j] += g[j];
    }
    t1 = wave_sum(t1) / (float)d;
    t2 = wave_sum(t2) / (float)d;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = rs2 * (g2[j] * g[j] - t1 - xh[j] * t2);
    st8g(dres_out + off, o);
    if (cast_out) {
      float m[8];
      drop_mask8(cast_drop, (uint32_t)row * (uint32_t)d + (uint32_t)(lane * 8), m);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] *= cast_scale * m[j];
      st8g(cast_out + off, o);
    }
  }
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    float* dgam = pass == 0 ? dgamma1 : dgamma2;
    float* dbet = pass == 0 ? dbeta1 : dbeta2;
    if (pass) __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[wave][0][lane * 8 + j] = pass == 0 ? ag1[j] : ag2[j];
      red[wave][1][lane * 8 + j] = pass == 0 ? ab1[j] : ab2[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 64 * LNB_WAVES) {
      float sg_ = 0.f, sb_ = 0.f;
#pragma unroll
      for (int w = 0; w < LNB_WAVES; ++w) { sg_ += red[w][0][c]; sb_ += red[w][1][c]; }
      atomicAdd(dgam + c, sg_);
      atomicAdd(dbet + c, sb_);
    }
  }
}
extern "C" int mi355x_layernorm2_bwd(const void* dy1, int dy1_dt, const void* x1, const void* gamma1, const void* mean1,
                                     const void* rstd1, void* dgamma1, void* dbeta1, const void* dres_in, const void* x2,
                                     const void* gamma2, const void* mean2, const void* rstd2, void* dgamma2, void* dbeta2,
                                     void* dres_out, int M, int d, void* cast_out, float cast_scale, unsigned drop_key,
                                     unsigned drop_threshold, float drop_scale, void* stream) {
  mi_clear_errors();
  if (!dy1 || !x1 || !gamma1 || !mean1 || !rstd1 || !dgamma1 || !dbeta1 || !dres_in || !x2 || !gamma2 || !mean2 || !rstd2 ||
      !dgamma2 || !dbeta2 || !dres_out || M <= 0 || d != 512)
    return MI_ERR_ARG;
  if (((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)dres_in | (uintptr_t)dres_out | (uintptr_t)gamma1 | (uintptr_t)gamma2) & 31) return MI_ERR_ARG;
  if (((uintptr_t)dy1 | (uintptr_t)cast_out) & 15) return MI_ERR_ARG;
  DropCfg dc = mi_drop(drop_key, drop_threshold, drop_scale);
  dim3 grid((M + LNB_ROWS - 1) / LNB_ROWS), block(64 * LNB_WAVES);
  DISPATCH_DT(dy1_dt, TDY, MI_LAUNCH((ln2_bwd_fused8_kernel<TDY>), grid, block, 0, (hipStream_t)stream, (const TDY*)dy1, (const float*)x1,
                                     (const float*)gamma1, (const float*)mean1, (const float*)rstd1, (float*)dgamma1, (float*)dbeta1,
                                     (const float*)dres_in, (const float*)x2, (const float*)gamma2, (const float*)mean2,
                                     (const float*)rstd2, (float*)dgamma2, (float*)dbeta2, (float*)dres_out, M, (bf16_t*)cast_out,
                                     cast_scale, dc));
  return mi_check_launch();
}

static int layernorm_bwd_impl(const void* dy, int dy_dt, const void* x, int x_dt, const void* gamma, const void* mean,
                              const void* rstd, void* dres, int accumulate, void* dgamma, void* dbeta, int M, int d,
                              void* cast_out, float cast_scale, DropCfg cast_drop, void* stream);
extern "C" int mi355x_layernorm_bwd(const void* dy, int dy_dt, const void* x, int x_dt, const void* gamma, const void* mean,
                                    const void* rstd, void* dres, int accumulate, void* dgamma, void* dbeta, int M, int d,
                                    void* stream) {
  mi_clear_errors();
  DropCfg nodrop = mi_drop(0u, 0u, 1.f);
  return layernorm_bwd_impl(dy, dy_dt, x, x_dt, gamma, mean, rstd, dres, accumulate, dgamma, dbeta, M, d, nullptr, 1.f, nodrop,
                            stream);
}
extern "C" int mi355x_layernorm_bwd_cast(const void* dy, int dy_dt, const void* x, int x_dt, const void* gamma, const void* mean,
                                         const void* rstd, void* dres, int accumulate, void* dgamma, void* dbeta, int M, int d,
                                         void* cast_out, float cast_scale, unsigned drop_key, unsigned drop_threshold,
                                         float drop_scale, void* stream) {
  mi_clear_errors();
  if (!cast_out || ((uintptr_t)cast_out & 15)) return MI_ERR_ARG;
  DropCfg dc = mi_drop(drop_key, drop_threshold, drop_scale);
  return layernorm_bwd_impl(dy, dy_dt, x, x_dt, gamma, mean, rstd, dres, accumulate, dgamma, dbeta, M, d, cast_out, cast_scale, dc,
                            stream);
}
__global__ __launch_bounds__(256) void ln_cast_after_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long long n8,
                                                            float alpha, DropCfg drop) {
  drop_resolve(drop);  // fallback: separate cast pass
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    float a[8], m[8];
    ld8g(in + i * 8, a);
    drop_mask8(drop, (uint32_t)(i * 8), m);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] *= alpha * m[j];
    st8g(out + i * 8, a);
  }
}
static int layernorm_bwd_impl(const void* dy, int dy_dt, const void* x, int x_dt, const void* gamma, const void* mean,
                              const void* rstd, void* dres, int accumulate, void* dgamma, void* dbeta, int M, int d,
                              void* cast_out, float cast_scale, DropCfg cast_drop, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dres || M <= 0 || d <= 0 || (d & 3)) return MI_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  dim3 block(256);
  const bool al = !((uintptr_t)dy & 15) && !((uintptr_t)x & 31) && !((uintptr_t)dres & 31) && !((uintptr_t)gamma & 31);
  if (al && (d == 512 || d == 1024) && ((dgamma && dbeta) || (!dgamma && !dbeta))) {
    dim3 gridf((M + LNB_ROWS - 1) / LNB_ROWS);
    dim3 blockf(64 * LNB_WAVES);
#define LN_F8(NCH) DISPATCH_DT(x_dt, TX, DISPATCH_DT(dy_dt, TDY, \
      MI_LAUNCH((ln_bwd_fused8_kernel<TX, TDY, NCH>), gridf, blockf, 0, s, (const TDY*)dy, (const TX*)x, \
                         (const float*)gamma, (const float*)mean, (const float*)rstd, (float*)dres, accumulate, \
                         (float*)dgamma, (float*)dbeta, M, d, (bf16_t*)cast_out, cast_scale, cast_drop)))
    if (d == 512) { LN_F8(1); } else { LN_F8(2); }
#undef LN_F8
    return mi_check_launch();
  }
  struct CastAfter {  // any other path: run the plain kernels, then the cast as its own pass
    void* out; float scale; DropCfg drop; const void* in; long long n; hipStream_t s;
    ~CastAfter() {
      if (out && (n & 7) == 0) {
        long long nb = (n / 8 + 255) / 256; if (nb > 4096) nb = 4096;
        MI_LAUNCH(ln_cast_after_kernel, dim3((unsigned)nb), dim3(256), 0, s, (const float*)in, (bf16_t*)out, n / 8,
                           scale, drop);
      }
    }
  } cast_after{cast_out, cast_scale, cast_drop, dres, (long long)M * d, s};
  if (cast_out && (((long long)M * d) & 7)) return MI_ERR_ARG;
  if (d <= 1024 && ((dgamma && dbeta) || (!dgamma && !dbeta))) {
    dim3 gridf((M + LNB_ROWS - 1) / LNB_ROWS);
    dim3 blockf(64 * LNB_WAVES);
    const int nvv = (d / 4 + 63) / 64;
#define LN_FUSED(NV) DISPATCH_DT(x_dt, TX, DISPATCH_DT(dy_dt, TDY, \
      MI_LAUNCH((ln_bwd_fused_kernel<TX, TDY, NV>), gridf, blockf, 0, s, (const TDY*)dy, (const TX*)x, \
                         (const float*)gamma, (const float*)mean, (const float*)rstd, (float*)dres, accumulate, \
                         (float*)dgamma, (float*)dbeta, M, d)))
    switch (nvv) {
      case 1: LN_FUSED(1); break;
      case 2: LN_FUSED(2); break;
      case 3: LN_FUSED(3); break;
      default: LN_FUSED(4); break;
    }
    return mi_check_launch();
  }
  if (dgamma && dbeta) {
    dim3 gp((d + 63) / 64, (M + CR_ROWS - 1) / CR_ROWS);
    DISPATCH_DT(x_dt, TX, DISPATCH_DT(dy_dt, TDY,
      MI_LAUNCH((ln_bwd_param_kernel<TX, TDY>), gp, block, 0, s, (const TDY*)dy, (const TX*)x, (const float*)mean,
                         (const float*)rstd, (float*)dgamma, (float*)dbeta, M, d)));
  }
  dim3 grid((M + 3) / 4);
  DISPATCH_DT(x_dt, TX, DISPATCH_DT(dy_dt, TDY,
    MI_LAUNCH((ln_bwd_dx_kernel<TX, TDY>), grid, block, 0, s, (const TDY*)dy, (const TX*)x, (const float*)gamma,
                       (const float*)mean, (const float*)rstd, (float*)dres, accumulate, M, d)));
  return mi_check_launch();
}

extern "C" int mi355x_colsum(const void* x, int x_dt, long long ld, void* out, int M, int N, float alpha, void* stream) {
  mi_clear_errors();
  if (!x || !out || M <= 0 || N <= 0) return MI_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (x_dt == MI_DT_BF16 && !(N & 7) && !(ld & 7) && !((uintptr_t)x & 15)) {
    dim3 grid((N + 511) / 512, (M + CSV_ROWS - 1) / CSV_ROWS);
    MI_LAUNCH(colsum_bf16x8_kernel, grid, dim3(256), 0, s, (const bf16_t*)x, ld, (float*)out, M, N, alpha);
    return mi_check_launch();
  }
  dim3 grid((N + 63) / 64, (M + CR_ROWS - 1) / CR_ROWS), block(256);
  DISPATCH_DT(x_dt, TX, MI_LAUNCH((colsum_kernel<TX>), grid, block, 0, s, (const TX*)x, ld, (float*)out, M, N, alpha));
  return mi_check_launch();
}

extern "C" int mi355x_log_softmax_fwd(const void* logits, long long ld_in, void* logp, long long ld_out, int M, int C,
                                      void* stream) {
  mi_clear_errors();
  if (!logits || !logp || M <= 0 || C <= 0) return MI_ERR_ARG;
  MI_LAUNCH(log_softmax_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)logits,
                     ld_in, (float*)logp, ld_out, M, C);
  return mi_check_launch();
}

extern "C" int mi355x_log_softmax_bwd(const void* dlogp, const void* logp, long long ld, void* dlogits, int out_dt,
                                      long long ld_out, int M, int C, float scale, void* stream) {
  mi_clear_errors();
  if (!dlogp || !logp || !dlogits || M <= 0 || C <= 0 || ld_out < C) return MI_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_DT(out_dt, TO, MI_LAUNCH((log_softmax_bwd_kernel<TO>), dim3((M + 3) / 4), dim3(256), 0, s,
                                             (const float*)dlogp, (const float*)logp, ld, (TO*)dlogits, ld_out, M, C, scale));
  return mi_check_launch();
}
