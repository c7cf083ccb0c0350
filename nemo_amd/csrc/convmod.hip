// Conformer convolution module, memory-bound part: depthwise Conv1d(k) along time on the [B,T,d] (channels-last)
// activation layout + training-mode BatchNorm1d statistics, BN-apply + Swish, and their backward passes.
//
// Replaces on the reference path (nemo/collections/asr/parts/submodules/conformer_modules.py:333-342):
//   CausalConv1D(d, d, k=31, groups=d, padding 15/15) (causal_convs.py:130-147) -> nn.BatchNorm1d -> Swish
// BN statistics are taken over all B*T positions (padded frames included) exactly as torch BatchNorm1d does on the
// reference's [B,d,T] tensor; sums are accumulated in f64 so SyncBN (all-reduce of the raw sums) is exact.
#include <stdlib.h>
#include "common.h"
#include "mi355x_asr.h"

#define DISPATCH_DT(dt, T, ...)                                      \
  if ((dt) == MI_DT_F32) { typedef float T; __VA_ARGS__; }           \
  else { typedef bf16_t T; __VA_ARGS__; }

#define DW_CH 64     // channels per block
#define DW_TT 64     // time steps per tile
#define DW_TQ 16     // outputs per thread (4 time groups x 16)
#define DW_MAXK 31
#define DW_LD (DW_CH + 4)  // LDS row pitch: +4 floats keeps float4 alignment and staggers rows across banks

__device__ __forceinline__ float round_as(float v, float) { return v; }
__device__ __forceinline__ float round_as(float v, bf16_t) { return bf2f(f2bf(v)); }

// 16-B (bf16: 8 channels) / 16-B (f32: 4 channels) global accesses: a tile row of 64 channels is 128 B (bf16) of contiguous
// memory; 2-byte-per-lane loads / stores run at a fraction of the HBM rate on this chip (measured on the first GEMM
// epilogue), so tiles move between HBM and LDS in vector chunks and are converted to f32 on the LDS side.
// rows [t_first, t_first + nrows) x channels [c0, c0+64) of x[b] -> tile[nrows][DW_CH] (f32), zero outside [0,T) x [0,d)
template <typename TT>
__device__ __forceinline__ void stage_tile(const TT* xb, int T, int d, int t_first, int nrows, int c0, float (*tile)[DW_LD]) {
  constexpr int V = VecIO<TT>::V, CPR = DW_CH / V;
  for (int q = threadIdx.x; q < nrows * CPR; q += 256) {
    const int r = q / CPR, cc = (q - r * CPR) * V;
    const int t = t_first + r, c = c0 + cc;
    float v[V];
    if (t >= 0 && t < T && c + V <= d) VecIO<TT>::load(xb + (long long)t * d + c, v);
    else {
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] = (t >= 0 && t < T && c + j < d) ? ld(xb + (long long)t * d + c + j) : 0.f;
    }
#pragma unroll
    for (int j = 0; j < V; j += 4) *reinterpret_cast<float4*>(&tile[r][cc + j]) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
  }
}
// tile[DW_TT][DW_CH] (f32) -> rows [t0, t0+DW_TT) x channels [c0, c0+64) of y[b]
template <typename TT>
__device__ __forceinline__ void unstage_tile(TT* yb, int T, int d, int t0, int c0, float (*tile)[DW_LD]) {
  constexpr int V = VecIO<TT>::V, CPR = DW_CH / V;
  for (int q = threadIdx.x; q < DW_TT * CPR; q += 256) {
    const int r = q / CPR, cc = (q - r * CPR) * V;
    const int t = t0 + r, c = c0 + cc;
    if (t >= T) continue;
    if (c + V <= d) VecIO<TT>::store(yb + (long long)t * d + c, &tile[r][cc]);
    else {
      for (int j = 0; j < V; ++j) if (c + j < d) st(yb + (long long)t * d + c + j, tile[r][cc + j]);
    }
  }
}

// BatchNorm + Swish backward applied on the way INTO the tile (the fused depthwise backward below): the tile receives
//   dcc = gamma * rstd * (dy * swish'(gamma * xh + beta) - k1 - xh * k2),  xh = (cc - mean) * rstd
// -- bn_swish_bwd_apply_kernel's arithmetic, rounded to the activation type exactly as that kernel's output tensor would be -- so the
// [M, d] gradient between the two kernels is never written or read.  coef: [6][DW_CH] = mean, rstd, gamma, beta, k1, k2 of the
// workgroup's channels.
template <typename TT>
__device__ __forceinline__ void stage_tile_bn(const TT* dyb, const TT* ccb, int T, int d, int t_first, int nrows, int c0,
                                              float (*tile)[DW_LD], const float (*coef)[DW_CH]) {
  constexpr int V = VecIO<TT>::V, CPR = DW_CH / V;
  for (int q = threadIdx.x; q < nrows * CPR; q += 256) {
    const int r = q / CPR, cc = (q - r * CPR) * V;
    const int t = t_first + r, c = c0 + cc;
    float v[V], u[V];
    const bool in = t >= 0 && t < T;
    if (in && c + V <= d) {
      VecIO<TT>::load(dyb + (long long)t * d + c, v);
      VecIO<TT>::load(ccb + (long long)t * d + c, u);
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const bool ok = in && c + j < d;
        v[j] = ok ? ld(dyb + (long long)t * d + c + j) : 0.f;
        u[j] = ok ? ld(ccb + (long long)t * d + c + j) : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float xh = (u[j] - coef[0][cc + j]) * coef[1][cc + j];
      float dz = v[j] * swish_grad(coef[2][cc + j] * xh + coef[3][cc + j]);
      dz -= coef[4][cc + j] + xh * coef[5][cc + j];
      v[j] = (in && c + j < d) ? round_as(coef[2][cc + j] * coef[1][cc + j] * dz, TT()) : 0.f;
    }
#pragma unroll
    for (int j = 0; j < V; j += 4) *reinterpret_cast<float4*>(&tile[r][cc + j]) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
  }
}
// tile[DW_TT][DW_CH] (f32: d(depthwise input) = d(GLU output)) -> the two halves of d(GLU input); see DwBnArgs
template <typename TT>
__device__ __forceinline__ void unstage_tile_glu(const TT* gin, TT* gdin, const long long* len, const long long* cu, int b, int T, int d,
                                                 int t0, int c0, float (*tile)[DW_LD], int act) {
  constexpr int V = VecIO<TT>::V, CPR = DW_CH / V;
  const int L = len ? (int)min((long long)T, len[b]) : T;
  const long long base = cu ? cu[b] : (long long)b * T;
  for (int q = threadIdx.x; q < DW_TT * CPR; q += 256) {
    const int r = q / CPR, cc = (q - r * CPR) * V;
    const int t = t0 + r, c = c0 + cc;
    if (t >= T || c >= d) continue;          // (d is a multiple of V: a chunk is inside or outside)
    const bool valid = t < L;
    if (cu && !valid) continue;
    const long long row = base + t;
    float da[V], dg[V];
    if (act == 1) {   // Swish backward of a [rows, d] matrix
      if (valid) {
        float a[V];
        VecIO<TT>::load(gin + row * d + c, a);
#pragma unroll
        for (int j = 0; j < V; ++j) da[j] = round_as(tile[r][cc + j], TT()) * swish_grad(a[j]);
      } else {
#pragma unroll
        for (int j = 0; j < V; ++j) da[j] = 0.f;
      }
      VecIO<TT>::store(gdin + row * d + c, da);
      continue;
    }
    if (valid) {
      float a[V], g[V];
      VecIO<TT>::load(gin + row * 2 * d + c, a);
      VecIO<TT>::load(gin + row * 2 * d + d + c, g);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float e = round_as(tile[r][cc + j], TT());   // (the two-launch form rounds dx to the activation type in between)
        const float sg = sigmoidf_(g[j]);
        da[j] = e * sg;
        dg[j] = e * a[j] * sg * (1.f - sg);
      }
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) { da[j] = 0.f; dg[j] = 0.f; }
    }
    VecIO<TT>::store(gdin + row * 2 * d + c, da);
    VecIO<TT>::store(gdin + row * 2 * d + d + c, dg);
  }
}
struct DwBnArgs {           // BatchNorm + Swish backward in front of the depthwise backward (null cc: plain depthwise backward)
  const void* cc;           // BatchNorm input (= the depthwise conv's output), same type and layout as dy
  const float *mean, *rstd, *gamma, *beta;
  const double* sums;       // f64 [2][d]: sum dz, sum dz * xhat over all rows (bn_swish_bwd_reduce)
  double inv_count;         // 1 / rows (host-known) ...
  const double* count_dev;  // ... or the row count as a device f64 (SyncBatchNorm over ragged ranks)
  int training;
  // GLU backward applied on the way OUT of the tile (optional; conformer_modules.py:333-335 backward: glu -> depthwise_conv): instead
  // of dx [B,T,d] the kernel writes d(pointwise_conv1 output) [rows, 2d] = (dx * sigmoid(b), dx * a * sigmoid'(b)), a | b = glu_in.
  // glu_cu (packed rows): glu_in / glu_din hold only the valid frames, utterance b at rows cu[b]..; else the padded grid b*T + t.
  // Frames beyond len[b] get zeros (padded grid) / have no row (packed).
  const void* glu_in;
  void* glu_din;
  const long long* glu_len;
  const long long* glu_cu;
  int glu_act;   // 0: GLU ([rows, 2d]), 1: Swish ([rows, d]; Squeezeformer's pointwise activation)
  int pad_shift;            // asymmetric padding: left pad = (KS - 1) / 2 + pad_shift (0 = symmetric; causal_convs.py:89-150)
};

// GLU (+ pad mask) applied on the way INTO the forward tile (conformer_modules.py:324-331 in front of the depthwise conv): the tile
// receives x = a * sigmoid(b) * (t < len[b]) from the pointwise conv's [rows, 2d] output (packed rows with cu, see glu_fwd_kernel),
// rounded to the activation type as the stand-alone GLU kernel's output would be; the workgroup's OWN rows [t0, t0 + DW_TT) are also
// written to gout [B,T,d] (the depthwise weight gradient's operand in backward) -- the halo rows belong to the neighbours.
template <typename TT>
__device__ __forceinline__ void stage_tile_glu(const TT* gin, TT* gout, const long long* len, const long long* cu, int b, int T, int d,
                                               int t_first, int nrows, int t0, int c0, float (*tile)[DW_LD], int act) {
  constexpr int V = VecIO<TT>::V, CPR = DW_CH / V;
  const int L = len ? (int)min((long long)T, len[b]) : T;
  const long long base = cu ? cu[b] : (long long)b * T;
  for (int q = threadIdx.x; q < nrows * CPR; q += 256) {
    const int r = q / CPR, cc = (q - r * CPR) * V;
    const int t = t_first + r, c = c0 + cc;
    float v[V];
#pragma unroll
    for (int j = 0; j < V; ++j) v[j] = 0.f;
    if (t >= 0 && t < L && c < d) {   // (d is a multiple of V: a chunk is inside or outside)
      float a[V], g[V];
      if (act == 0) {   // GLU: a | b halves of a [rows, 2d] matrix
        VecIO<TT>::load(gin + (base + t) * 2 * d + c, a);
        VecIO<TT>::load(gin + (base + t) * 2 * d + d + c, g);
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = round_as(a[j] * sigmoidf_(g[j]), TT());
      } else {          // Swish of a [rows, d] matrix (Squeezeformer's conv module: pointwise_activation = 'swish')
        VecIO<TT>::load(gin + (base + t) * d + c, a);
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = round_as(swishf_(a[j]), TT());
      }
    }
#pragma unroll
    for (int j = 0; j < V; j += 4) *reinterpret_cast<float4*>(&tile[r][cc + j]) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
    if (t >= t0 && t < t0 + DW_TT && t < T && c < d) VecIO<TT>::store(gout + ((long long)b * T + t) * d + c, v);
  }
}
struct DwGluArgs {   // GLU in front of the depthwise forward (null in: plain depthwise forward)
  const void* in;    // [rows, 2d]
  void* out;         // [B, T, d] GLU output (kept for backward)
  const long long* len;
  const long long* cu;
  int act;           // 0: GLU of [rows, 2d], 1: Swish of [rows, d]
  int pad_shift;     // asymmetric padding: left pad = (KS - 1) / 2 + pad_shift (0 = symmetric; (KS - 1) / 2 = causal)
};

// ------------------------------------------------------------------------------------------------ forward
// x [B,T,d] -> y[b,t,c] = bias[c] + sum_k w[c,k] * x[b, t+k-pad, c]  (zero outside [0,T));  stats[0][c] += sum y, stats[1][c] += sum y^2
template <typename TT, int KS, bool GLU>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const TT* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, TT* __restrict__ y,
                                                         double* __restrict__ stats, int B, int T, int d, DwGluArgs glu) {
  const int PAD = (KS - 1) / 2 + glu.pad_shift;   // left pad (CausalConv1D: conv_context_size = [left, right], left + right + 1 = KS)
  constexpr int ROWS = DW_TT + KS - 1;
  __shared__ __attribute__((aligned(16))) float tile[ROWS][DW_LD];
  __shared__ __attribute__((aligned(16))) float otile[DW_TT][DW_LD];
  __shared__ float red[2][4][DW_CH];
  const int c_l = threadIdx.x & 63, tg = threadIdx.x >> 6;
  const int c0 = blockIdx.x * DW_CH;
  const int c = c0 + c_l;
  const int b = blockIdx.z;
  const int t0 = blockIdx.y * DW_TT;
  const bool cv = c < d;
  if (GLU) stage_tile_glu<TT>((const TT*)glu.in, (TT*)glu.out, glu.len, glu.cu, b, T, d, t0 - PAD, ROWS, t0, c0, tile, glu.act);
  else stage_tile<TT>(x + (long long)b * T * d, T, d, t0 - PAD, ROWS, c0, tile);
  float wk[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) wk[k] = cv ? w[c * KS + k] : 0.f;
  const float bs = (cv && bias) ? bias[c] : 0.f;
  __syncthreads();
  float in[DW_TQ + KS - 1];
#pragma unroll
  for (int i = 0; i < DW_TQ + KS - 1; ++i) in[i] = tile[tg * DW_TQ + i][c_l];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int o = 0; o < DW_TQ; ++o) {
    float a = bs;
#pragma unroll
    for (int k = 0; k < KS; ++k) a = fmaf(wk[k], in[o + k], a);
    const int t = t0 + tg * DW_TQ + o;
    otile[tg * DW_TQ + o][c_l] = a;
    if (cv && t < T) {
      const float ar = round_as(a, TT());
      s1 += ar; s2 += ar * ar;
    }
  }
  if (stats) { red[0][tg][c_l] = s1; red[1][tg][c_l] = s2; }
  __syncthreads();
  unstage_tile<TT>(y + (long long)b * T * d, T, d, t0, c0, otile);
  if (stats && tg == 0 && cv) {
    atomicAdd(stats + c, (double)((red[0][0][c_l] + red[0][1][c_l]) + (red[0][2][c_l] + red[0][3][c_l])));
    atomicAdd(stats + d + c, (double)((red[1][0][c_l] + red[1][1][c_l]) + (red[1][2][c_l] + red[1][3][c_l])));
  }
}

// ------------------------------------------------------------------------------------------------ backward
// dy [B,T,d], x [B,T,d]:  dx[t] = sum_k w[k] * dy[t + pad - k] ;  dw[c,k] += sum_{b,t} dy[t] * x[t+k-pad] ; dbias[c] += sum dy
// grid (d/64, DW_SEG, B): one block walks the time tiles of its segment, one atomic per (c,k) per block at the end.
#define DW_SEG 4
template <typename TT, int KS, bool BN, bool ASYM = false>
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(const TT* __restrict__ dy, const TT* __restrict__ x,
                                                         const float* __restrict__ w, TT* __restrict__ dx,
                                                         float* __restrict__ dw, float* __restrict__ dbias, float* __restrict__ partial, int B, int T, int d,
                                                         DwBnArgs bn) {
  // forward pads (left, right): y[t] = sum_k w[k] x[t + k - PADL].  The symmetric case keeps them compile-time constants (a run-time
  // pad costs the depthwise backward 57 -> 88 us per launch: the dy value of the tap gradients then comes from LDS, not a register)
  const int PADL = (KS - 1) / 2 + (ASYM ? bn.pad_shift : 0), PADR = KS - 1 - PADL;
  constexpr int ROWS = DW_TT + KS - 1;
  __shared__ __attribute__((aligned(16))) float big[2][ROWS][DW_LD];  // dy tile | x tile (also the final reduction buffer)
  float (*tdy)[DW_LD] = big[0];
  float (*tx)[DW_LD] = big[1];
  __shared__ __attribute__((aligned(16))) float otile[DW_TT][DW_LD];
  static_assert(4 * (KS + 1) * DW_CH <= 2 * ROWS * DW_CH, "reduction buffer must fit in the tile storage");
  const int c_l = threadIdx.x & 63, tg = threadIdx.x >> 6;
  const int c0 = blockIdx.x * DW_CH;
  const int c = c0 + c_l;
  const int b = blockIdx.z;
  const bool cv = c < d;
  const int ntile = (T + DW_TT - 1) / DW_TT;
  const int per = (ntile + DW_SEG - 1) / DW_SEG;
  const int tile_lo = blockIdx.y * per, tile_hi = min(ntile, tile_lo + per);
  float wk[KS], gw[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) { wk[k] = cv ? w[c * KS + k] : 0.f; gw[k] = 0.f; }
  float gb = 0.f;
  __shared__ float coef[BN ? 6 : 1][DW_CH];
  if (BN && threadIdx.x < DW_CH) {   // (visible behind the first barrier of the tile loop)
    const int cl = threadIdx.x, cg = c0 + cl;
    const bool ok = cg < d;
    const double inv_count = bn.count_dev ? 1.0 / *bn.count_dev : bn.inv_count;
    coef[0][cl] = ok ? bn.mean[cg] : 0.f;
    coef[1][cl] = ok ? bn.rstd[cg] : 0.f;
    coef[2][cl] = ok ? bn.gamma[cg] : 0.f;
    coef[3][cl] = ok ? bn.beta[cg] : 0.f;
    coef[4][cl] = (ok && bn.training) ? (float)(bn.sums[cg] * inv_count) : 0.f;
    coef[5][cl] = (ok && bn.training) ? (float)(bn.sums[d + cg] * inv_count) : 0.f;
  }
  for (int ti = tile_lo; ti < tile_hi; ++ti) {
    const int t0 = ti * DW_TT;
    __syncthreads();
    // dx[t] = sum_k w[k] dy[t + PADL - k]: the dy window starts PADR frames before the tile; dw[k] += dy[t] x[t + k - PADL]: the x
    // window PADL frames before it
    if (BN) stage_tile_bn<TT>(dy + (long long)b * T * d, (const TT*)bn.cc + (long long)b * T * d, T, d, t0 - PADR, ROWS, c0, tdy, coef);
    else stage_tile<TT>(dy + (long long)b * T * d, T, d, t0 - PADR, ROWS, c0, tdy);
    stage_tile<TT>(x + (long long)b * T * d, T, d, t0 - PADL, ROWS, c0, tx);
    __syncthreads();
    float vdy[DW_TQ + KS - 1], vx[DW_TQ + KS - 1];
#pragma unroll
    for (int i = 0; i < DW_TQ + KS - 1; ++i) { vdy[i] = tdy[tg * DW_TQ + i][c_l]; vx[i] = tx[tg * DW_TQ + i][c_l]; }
#pragma unroll
    for (int o = 0; o < DW_TQ; ++o) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < KS; ++k) a = fmaf(wk[k], vdy[o + KS - 1 - k], a);
      otile[tg * DW_TQ + o][c_l] = a;
      // dy[t0 + o] (zero when t >= T: staging zero-fills); with a run-time pad from LDS
      const float g = ASYM ? tdy[tg * DW_TQ + o + PADR][c_l] : vdy[o + (KS - 1) / 2];

      gb += g;
#pragma unroll
      for (int k = 0; k < KS; ++k) gw[k] = fmaf(g, vx[o + k], gw[k]);
    }
    __syncthreads();
    if (BN && bn.glu_in) unstage_tile_glu<TT>((const TT*)bn.glu_in, (TT*)bn.glu_din, bn.glu_len, bn.glu_cu, b, T, d, t0, c0, otile, bn.glu_act);
    else unstage_tile<TT>(dx + (long long)b * T * d, T, d, t0, c0, otile);
  }
  // one LDS round for all KS+1 partial sums: [4 time groups][KS+1][64 channels] (re-uses the dy tile), then 2 atomics/thread
  __syncthreads();
  float* rbuf = &big[0][0][0];
#pragma unroll
  for (int k = 0; k < KS; ++k) rbuf[(tg * (KS + 1) + k) * DW_CH + c_l] = gw[k];
  rbuf[(tg * (KS + 1) + KS) * DW_CH + c_l] = gb;
  __syncthreads();
  // Same-address float atomics from 128 workgroups cost 90 us here (measured); with a scratch buffer every workgroup
  // stores its [KS+1][64] partial sums and reduce_partials_kernel adds them up (deterministic as a bonus).
  float* pslab = partial ? partial + ((long long)(blockIdx.z * DW_SEG + blockIdx.y) * (KS + 1)) * d : nullptr;
  for (int e = threadIdx.x; e < (KS + 1) * DW_CH; e += 256) {
    const int k = e / DW_CH, cl = e - k * DW_CH;
    const int cc = c0 + cl;
    if (cc >= d) continue;
    const float v = (rbuf[(0 * (KS + 1) + k) * DW_CH + cl] + rbuf[(1 * (KS + 1) + k) * DW_CH + cl]) +
                    (rbuf[(2 * (KS + 1) + k) * DW_CH + cl] + rbuf[(3 * (KS + 1) + k) * DW_CH + cl]);
    if (pslab) pslab[(long long)k * d + cc] = v;
    else if (k < KS) atomicAdd(dw + cc * KS + k, v);
    else if (dbias) atomicAdd(dbias + cc, v);
  }
}

// ------------------------------------------------------------------------------------------------ streaming depthwise conv (bf16)
// The tile kernels above move every activation through an f32 LDS image twice (stage -> barrier -> taps -> image -> barrier ->
// store) and ran at 1.35 / 1.5 TB/s for 33 / 50 MB of traffic: 27.7 / 51 us per layer (profiles/r3_pmc_hbm_traffic.md), bound by
// that per-workgroup chain, not by HBM or the vector unit (two rewrites of the SAME structure changed nothing, r3_raw/
// dwconv_packed_fma_experiment.txt).  Here the activations never touch LDS: a lane owns TWO adjacent channels (one packed-bf16
// dword per row), a wave 128 channels; a lane's window of DS_TQ + KS - 1 rows comes straight from global memory with dword loads
// (256 contiguous bytes per wave and row; the 2.9-fold re-read of neighbouring windows is L2 traffic) and its DS_TQ x 2 results
// go straight back as packed dwords.  No barrier between load and store; only the 31 x 128 weights of the workgroup's channel
// group are staged through LDS once (contiguous in memory, strided per lane).  The vector unit is the floor: 2 x 31 FMAs per
// output pair.  dx of the backward pass IS this kernel with the taps flipped; the weight gradient is a second streaming kernel.
#define DS_TQ 16                 // outputs per pass and lane (window = DS_TQ + KS - 1 rows)
#define DS_WT (2 * DS_TQ)        // outputs per wave: two passes over one 2*DS_TQ + KS - 1 row window
#define DS_NW 8                  // waves per workgroup of the forward / dx kernel (consecutive time tiles)
#define DS_BT (DS_NW * DS_WT)    // outputs per workgroup: 256
#define DS_CG 128                // channels per workgroup

typedef float ds_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float bf_lo(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf_hi(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }
// acc += w * x on both channels of the lane: one v_pk_fma_f32.  A packed op's result may not be read by the very next instruction
// (gfx940+ forwarding hazard: one wait state); hipcc orders every output's 31-tap chain back to back and pays an s_nop per FMA.
// Issued from here in tap-major order the dependent op is DS_TQ instructions away, so no wait state is ever needed.
__device__ __forceinline__ void ds_pk_fma(ds_v2f& acc, const ds_v2f& w, const ds_v2f& x) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x));
}

template <int KS, int NT>
__device__ __forceinline__ void ds_stage_weights(const float* __restrict__ w, int cg0, int d, float* wl) {
  for (int e = threadIdx.x; e < DS_CG * KS; e += NT) wl[e] = (cg0 * KS + e < d * KS) ? w[cg0 * KS + e] : 0.f;
  __syncthreads();
}
// the lane's channel pair of row t of one utterance.  EDGE: rows outside [0, T) read as zero (row index clamped, value selected:
// no branch); interior windows (all but the first and last tile of an utterance) load unconditionally from a running row pointer.
template <bool EDGE>
__device__ __forceinline__ ds_v2f ds_row(const char* __restrict__ ub, int t, int T, long long row_bytes, uint32_t lane_off, bool cv) {
  ds_v2f r;
  if (EDGE) {
    const int tc = min(max(t, 0), T - 1);
    const uint32_t v = *reinterpret_cast<const uint32_t*>(ub + (long long)tc * row_bytes + lane_off);
    const bool ok = cv && t >= 0 && t < T;
    r.x = ok ? bf_lo(v) : 0.f;
    r.y = ok ? bf_hi(v) : 0.f;
  } else {
    const uint32_t v = *reinterpret_cast<const uint32_t*>(ub + (long long)t * row_bytes + lane_off);
    r.x = bf_lo(v);
    r.y = bf_hi(v);
  }
  return r;
}

// one wave's DS_WT outputs of the forward / dx kernel
template <int KS, bool STATS, bool EDGE, int D>
__device__ __forceinline__ void ds_wave_tile(const char* __restrict__ ub, char* __restrict__ yb, int t0, int T, int d, uint32_t lane_off,
                                             bool cv, const ds_v2f (&wp)[KS], ds_v2f bv, ds_v2f& s1, ds_v2f& s2) {
  constexpr int PAD = (KS - 1) / 2, WIN2 = DS_WT + KS - 1;
  const long long row_bytes = D ? 2LL * D : 2LL * d;
  const char* wb = EDGE ? ub : ub + (long long)(t0 - PAD) * row_bytes;  // (interior: constant row offsets from the window's first row)
  ds_v2f xw[WIN2];
#pragma unroll
  for (int i = 0; i < WIN2; ++i) xw[i] = EDGE ? ds_row<true>(ub, t0 - PAD + i, T, row_bytes, lane_off, cv)
                                              : ds_row<false>(wb, i, T, row_bytes, lane_off, cv);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (EDGE && t0 + p * DS_TQ >= T) break;
    ds_v2f a[DS_TQ];
#pragma unroll
    for (int o = 0; o < DS_TQ; ++o) a[o] = bv;
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
      for (int o = 0; o < DS_TQ; ++o) ds_pk_fma(a[o], wp[k], xw[p * DS_TQ + o + k]);
#pragma unroll
    for (int o = 0; o < DS_TQ; ++o) {
      const int t = t0 + p * DS_TQ + o;
      if (!EDGE || (cv && t < T)) {
        const uint32_t pk = pack_bf2(a[o].x, a[o].y);
        *reinterpret_cast<uint32_t*>(yb + (long long)t * row_bytes + lane_off) = pk;
        if (STATS) {
          const float r0 = bf_lo(pk), r1 = bf_hi(pk);
          s1.x += r0; s2.x = fmaf(r0, r0, s2.x);
          s1.y += r1; s2.y = fmaf(r1, r1, s2.y);
        }
      }
    }
  }
}

// y[b,t,c] = bias[c] + sum_k w[c, FLIP ? KS-1-k : k] * x[b, t+k-PAD, c];  STATS: stats[0][c] += sum y, stats[1][c] += sum y^2 (of the
// bf16-rounded outputs, t < T).  D: the channel count at compile time (row offsets become instruction immediates), 0 = run time.
template <int KS, bool FLIP, bool STATS, int D>
__global__ __launch_bounds__(64 * DS_NW) void dwconv_stream_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                                   const float* __restrict__ bias, bf16_t* __restrict__ y,
                                                                   double* __restrict__ stats, int T, int d) {
  constexpr int PAD = (KS - 1) / 2;
  __shared__ float wl[DS_CG * KS];
  __shared__ float red[DS_NW][4][64];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cg0 = blockIdx.x * DS_CG;
  const int c = cg0 + 2 * lane;
  const bool cv = c < d;  // (d is even: both channels of the pair are valid or none)
  const uint32_t lane_off = (uint32_t)(cv ? c : 0) * 2u;
  const int b = blockIdx.z;
  ds_stage_weights<KS, 64 * DS_NW>(w, cg0, d, wl);
  ds_v2f wp[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    const int kk = FLIP ? KS - 1 - k : k;
    wp[k].x = wl[(2 * lane) * KS + kk];
    wp[k].y = wl[(2 * lane + 1) * KS + kk];
  }
  ds_v2f bv;
  bv.x = (cv && bias) ? bias[c] : 0.f;
  bv.y = (cv && bias) ? bias[c + 1] : 0.f;
  const long long row_bytes = D ? 2LL * D : 2LL * d;
  const char* ub = reinterpret_cast<const char*>(x) + (long long)b * T * row_bytes;
  char* yb = reinterpret_cast<char*>(y) + (long long)b * T * row_bytes;
  ds_v2f s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
  const int t0 = blockIdx.y * DS_BT + wv * DS_WT;
  if (t0 < T) {  // (wave-uniform)
    const bool interior = t0 - PAD >= 0 && t0 + DS_WT + PAD <= T && cg0 + DS_CG <= d;
    if (interior) ds_wave_tile<KS, STATS, false, D>(ub, yb, t0, T, d, lane_off, cv, wp, bv, s1, s2);
    else ds_wave_tile<KS, STATS, true, D>(ub, yb, t0, T, d, lane_off, cv, wp, bv, s1, s2);
  }
  if (STATS) {
    red[wv][0][lane] = s1.x; red[wv][1][lane] = s1.y; red[wv][2][lane] = s2.x; red[wv][3][lane] = s2.y;
    __syncthreads();
    // thread -> (quantity q, lane): sum over the waves, one f64 atomic per (channel, statistic) and workgroup
    if (wv < 4) {
      const int q = wv;
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < DS_NW; ++i) v += red[i][q][lane];
      const int cc = cg0 + 2 * lane + (q & 1);
      if (cc < d) atomicAdd(stats + (q >> 1) * d + cc, (double)v);
    }
  }
}

// weight / bias gradient partials: partial[p][k][c] = sum over the slab's outputs of dy[t,c] * x[t+k-PAD,c] (k < KS), [KS][c] = sum dy
// grid (d/128, DW_SEG, B): workgroup (cg, seg, b) walks the time tiles seg, seg + DW_SEG, ... of batch b (slab p = b*DW_SEG + seg);
// a wave's tile = DS_WT outputs in passes of DSW_TQ over a CIRCULAR window of DSW_TQ + KS - 1 rows (row r lives in slot r % WIN: a
// pass's new rows take the slots of the oldest -- all indices are compile-time, no register moves)
#define DSW_BT (4 * DS_WT)
#define DSW_TQ 8   // outputs per pass of the weight-gradient kernel (window = 38 rows: registers for three waves per SIMD)
template <int KS, bool EDGE, int D>
__device__ __forceinline__ void ds_wave_dw(const char* __restrict__ xb, const char* __restrict__ gb, int t0, int T, int d, uint32_t lane_off,
                                           bool cv, ds_v2f (&gw)[KS], ds_v2f& gbias) {
  constexpr int PAD = (KS - 1) / 2, WIN = DSW_TQ + KS - 1;
  const long long row_bytes = D ? 2LL * D : 2LL * d;
  const char* wb = EDGE ? xb : xb + (long long)(t0 - PAD) * row_bytes;
  const char* wg = EDGE ? gb : gb + (long long)t0 * row_bytes;
  ds_v2f xw[WIN], g[DSW_TQ];
#pragma unroll
  for (int i = 0; i < WIN; ++i) xw[i] = EDGE ? ds_row<true>(xb, t0 - PAD + i, T, row_bytes, lane_off, cv)
                                             : ds_row<false>(wb, i, T, row_bytes, lane_off, cv);
#pragma unroll
  for (int p = 0; p < DS_WT / DSW_TQ; ++p) {
    if (EDGE && t0 + p * DSW_TQ >= T) break;
#pragma unroll
    for (int o = 0; o < DSW_TQ; ++o) g[o] = EDGE ? ds_row<true>(gb, t0 + p * DSW_TQ + o, T, row_bytes, lane_off, cv)
                                                 : ds_row<false>(wg, p * DSW_TQ + o, T, row_bytes, lane_off, cv);
    if (p > 0) {
#pragma unroll
      for (int j = 0; j < DSW_TQ; ++j) {  // the pass's 8 new rows take the slots of the 8 oldest
        const int r = WIN + (p - 1) * DSW_TQ + j;
        xw[r % WIN] = EDGE ? ds_row<true>(xb, t0 - PAD + r, T, row_bytes, lane_off, cv) : ds_row<false>(wb, r, T, row_bytes, lane_off, cv);
      }
    }
#pragma unroll
    for (int o = 0; o < DSW_TQ; ++o) {
      gbias += g[o];
#pragma unroll
      for (int k = 0; k < KS; ++k) ds_pk_fma(gw[k], g[o], xw[(p * DSW_TQ + o + k) % WIN]);
    }
  }
}
template <int KS, int D>
__global__ __launch_bounds__(256) void dwconv_dw_stream_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                               float* __restrict__ partial, int T, int d) {
  constexpr int PAD = (KS - 1) / 2;
  extern __shared__ float slab[];  // [4 waves][KS + 1][128]
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cg0 = blockIdx.x * DS_CG;
  const int c = cg0 + 2 * lane;
  const bool cv = c < d;
  const uint32_t lane_off = (uint32_t)(cv ? c : 0) * 2u;
  const int b = blockIdx.z;
  const long long row_bytes = D ? 2LL * D : 2LL * d;
  const char* xb = reinterpret_cast<const char*>(x) + (long long)b * T * row_bytes;
  const char* gb = reinterpret_cast<const char*>(dy) + (long long)b * T * row_bytes;
  ds_v2f gw[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) gw[k] = ds_v2f{0.f, 0.f};
  ds_v2f gbias = {0.f, 0.f};
  const int ntile = (T + DSW_BT - 1) / DSW_BT;
  for (int ti = blockIdx.y; ti < ntile; ti += gridDim.y) {
    const int t0 = ti * DSW_BT + wv * DS_WT;
    if (t0 >= T) break;
    const bool interior = t0 - PAD >= 0 && t0 + DS_WT + PAD <= T && cg0 + DS_CG <= d;
    if (interior) ds_wave_dw<KS, false, D>(xb, gb, t0, T, d, lane_off, cv, gw, gbias);
    else ds_wave_dw<KS, true, D>(xb, gb, t0, T, d, lane_off, cv, gw, gbias);
  }
  float* mine = slab + wv * (KS + 1) * DS_CG;
#pragma unroll
  for (int k = 0; k < KS; ++k) *reinterpret_cast<ds_v2f*>(mine + k * DS_CG + 2 * lane) = gw[k];
  *reinterpret_cast<ds_v2f*>(mine + KS * DS_CG + 2 * lane) = gbias;
  __syncthreads();
  float* pslab = partial + ((long long)(b * gridDim.y + blockIdx.y) * (KS + 1)) * d;
  constexpr int PS = (KS + 1) * DS_CG;
  for (int e = threadIdx.x; e < PS; e += 256) {
    const int k = e / DS_CG, cl = e - k * DS_CG;
    if (cg0 + cl >= d) continue;
    pslab[(long long)k * d + cg0 + cl] = (slab[e] + slab[PS + e]) + (slab[2 * PS + e] + slab[3 * PS + e]);
  }
}

// ------------------------------------------------------------------------------------------------ BatchNorm
// 1 / sqrt(var + eps) in f32 with one Newton step on the hardware estimate (<= 1 ulp; the mean and the variance themselves come
// from the f64 sums): shared by the stand-alone and the fused statistics kernels so that both give the same bits
__device__ __forceinline__ float bn_rstd(float var, float eps) {
  const float x = var + eps;
  float r = rsqrtf(x);
  r = r * (1.5f - 0.5f * x * r * r);
  return r;
}
// stats (f64 [2][d]: sum, sum of squares over `count` positions) -> mean, rstd (biased var), running stats update
// `count_dev` (optional): the element count as a device f64 -- under SyncBatchNorm it is the (all-reduced) sum of the ranks'
// own B*T' and travels in the same buffer as the sums, so ragged ranks need no host round trip
__global__ void bn_finalize_kernel(const double* __restrict__ stats, double count, const double* __restrict__ count_dev,
                                   float* __restrict__ mean,
                                   float* __restrict__ rstd, float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float momentum, float eps, int d) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d) return;
  if (count_dev) count = *count_dev;
  const double inv_count = 1.0 / count;
  const double mu = stats[c] * inv_count;
  double var = stats[d + c] * inv_count - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)mu;
  rstd[c] = bn_rstd((float)var, eps);
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
  if (running_var) {
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}
// eval-mode helper: rstd = 1/sqrt(running_var + eps), mean = running_mean
__global__ void bn_eval_stats_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                     float* __restrict__ mean, float* __restrict__ rstd, float eps, int d) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d) return;
  mean[c] = running_mean[c];
  rstd[c] = rsqrtf(running_var[c] + eps);
}

// y = swish(gamma * (x - mean) * rstd + beta)
// thread = V consecutive channels (fixed for the whole kernel: the per-channel coefficients are loaded once) x a strided
// set of rows
template <typename TT>
__global__ __launch_bounds__(256) void bn_swish_fwd_kernel(const TT* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, TT* __restrict__ y, long long M, int d) {
  constexpr int V = VecIO<TT>::V;
  const int CP = min(d / V, 256), RS = 256 / CP;
  const int ck = threadIdx.x % CP, rsub = threadIdx.x / CP;
  if (rsub >= RS) return;
  for (int c = ck * V; c < d; c += CP * V) {
    float mu[V], rs[V], g[V], bt[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { mu[j] = mean[c + j]; rs[j] = rstd[c + j]; g[j] = gamma[c + j]; bt[j] = beta[c + j]; }
    for (long long m = (long long)blockIdx.x * RS + rsub; m < M; m += (long long)gridDim.x * RS) {
      float v[V], o[V];
      VecIO<TT>::load(x + m * d + c, v);
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = swishf_(g[j] * (v[j] - mu[j]) * rs[j] + bt[j]);
      VecIO<TT>::store(y + m * d + c, o);
    }
  }
}
// The same with the statistics finalised in the kernel (training forward): every thread derives mean / rstd of ITS channels from
// the f64 sums (bn_finalize_kernel's arithmetic), workgroup 0 also writes them out for backward and updates the running
// statistics -- one launch per layer instead of two.
template <typename TT>
__global__ __launch_bounds__(256) void bn_stats_swish_fwd_kernel(const TT* __restrict__ x, const double* __restrict__ stats,
                                                                 double count, double inv_count_host,
                                                                 const double* __restrict__ count_dev,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 TT* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
                                                                 float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                 float momentum, float eps, long long M, int d) {
  constexpr int V = VecIO<TT>::V;
  const int CP = min(d / V, 256), RS = 256 / CP;
  const int ck = threadIdx.x % CP, rsub = threadIdx.x / CP;
  if (rsub >= RS) return;
  if (count_dev) count = *count_dev;
  const double inv_count = count_dev ? 1.0 / count : inv_count_host;  // (host-known count: its reciprocal comes as an argument)
  for (int c = ck * V; c < d; c += CP * V) {
    float mu[V], rs[V], g[V], bt[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      // (at most one f64 reciprocal per thread, then multiplies: EVERY workgroup derives the coefficients of its channels, and f64
      //  divisions / square roots here cost 20 us per launch -- more than the whole normalisation pass)
      const double m_ = stats[c + j] * inv_count;
      double var = stats[d + c + j] * inv_count - m_ * m_;
      if (var < 0.0) var = 0.0;
      mu[j] = (float)m_; rs[j] = bn_rstd((float)var, eps);
      g[j] = gamma[c + j]; bt[j] = beta[c + j];
      if (blockIdx.x == 0 && rsub == 0) {
        mean[c + j] = mu[j]; rstd[c + j] = rs[j];
        if (running_mean) running_mean[c + j] = (1.f - momentum) * running_mean[c + j] + momentum * mu[j];
        if (running_var) {
          const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
          running_var[c + j] = (1.f - momentum) * running_var[c + j] + momentum * (float)unb;
        }
      }
    }
    for (long long m = (long long)blockIdx.x * RS + rsub; m < M; m += (long long)gridDim.x * RS) {
      float v[V], o[V];
      VecIO<TT>::load(x + m * d + c, v);
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = swishf_(g[j] * (v[j] - mu[j]) * rs[j] + bt[j]);
      VecIO<TT>::store(y + m * d + c, o);
    }
  }
}
#define BNR_ROWS 32  // rows per workgroup (501 workgroups at the Large shape; partial sums go through a scratch slab)
template <typename TT>
__global__ __launch_bounds__(256) void bn_swish_bwd_reduce_kernel(const TT* __restrict__ dy, const TT* __restrict__ x,
                                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  double* __restrict__ sums, float* __restrict__ partial,
                                                                  long long M, int d, int rows_per_blk) {
  // thread = V consecutive channels (one 16-byte load per tensor per row) x every RS-th row of the workgroup's row block
  constexpr int V = VecIO<TT>::V;
  // ONE reduction slab (8 KiB for bf16), used for the two sums one after the other: with two slabs (16 KiB) this kernel missed
  // sharing a CU with the 144-KiB weight-gradient workgroups of the side stream by exactly 1 KiB (LDS of co-resident workgroups must
  // sum to < 160 KiB: tools/probe/coreside_probe.hip) and ran 94 us instead of 13.5 us inside the step
  __shared__ float sred[256 * V];
  const int CP = min(d / V, 256);          // channel chunks per pass
  const int RS = 256 / CP;                 // rows in flight per pass
  const int ck = threadIdx.x % CP, rsub = threadIdx.x / CP;
  const long long r0 = (long long)blockIdx.y * rows_per_blk, r1 = min(M, r0 + rows_per_blk);
  for (int c0 = blockIdx.x * CP * V; c0 < d; c0 += gridDim.x * CP * V) {
    const int c = c0 + ck * V;
    float a1[V], a2[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { a1[j] = 0.f; a2[j] = 0.f; }
    if (rsub < RS && c < d) {
      float mu[V], rs[V], g[V], bt[V];
#pragma unroll
      for (int j = 0; j < V; ++j) { mu[j] = mean[c + j]; rs[j] = rstd[c + j]; g[j] = gamma[c + j]; bt[j] = beta[c + j]; }
#pragma unroll 3   // (4 rows in flight need 162 VGPRs; beside two 176-register weight-gradient waves a SIMD has 160 left)
      for (long long r = r0 + rsub; r < r1; r += RS) {
        float xv[V], dv[V];
        VecIO<TT>::load(x + r * d + c, xv);
        VecIO<TT>::load(dy + r * d + c, dv);
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const float xh = (xv[j] - mu[j]) * rs[j];
          const float dz = dv[j] * swish_grad(g[j] * xh + bt[j]);
          a1[j] += dz; a2[j] += dz * xh;
        }
      }
    }
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < V; ++j) sred[threadIdx.x * V + j] = which ? a2[j] : a1[j];
      __syncthreads();
      for (int e = threadIdx.x; e < CP * V; e += 256) {  // e = channel inside this pass
        float t = 0.f;
        for (int q = 0; q < RS; ++q) t += sred[q * CP * V + e];
        if (c0 + e < d) {
          if (partial) partial[((long long)blockIdx.y * 2 + which) * d + c0 + e] = t;
          else atomicAdd(sums + which * d + c0 + e, (double)t);
        }
      }
    }
  }
}
template <typename TT, int UNR>
__global__ __launch_bounds__(256) void bn_swish_bwd_apply_kernel(const TT* __restrict__ dy, const TT* __restrict__ x,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const double* __restrict__ sums, double inv_count_host,
                                                                 const double* __restrict__ count_dev, int training,
                                                                 TT* __restrict__ dx, long long M, int d) {
  // per-channel coefficients are built once per workgroup in LDS (the f64 mean-of-sums included), so a thread's set-up is
  // six 16/32-byte LDS reads however few rows it handles
  constexpr int V = VecIO<TT>::V;
  extern __shared__ float coef[];  // [6][d]: mean, rstd, gamma, beta, k1 = sum(dz)/n, k2 = sum(dz*xhat)/n
  const double inv_count = count_dev ? 1.0 / *count_dev : inv_count_host;  // (host-known count: no f64 division per thread)
  for (int c = threadIdx.x; c < d; c += 256) {
    coef[c] = mean[c]; coef[d + c] = rstd[c]; coef[2 * d + c] = gamma[c]; coef[3 * d + c] = beta[c];
    coef[4 * d + c] = training ? (float)(sums[c] * inv_count) : 0.f;
    coef[5 * d + c] = training ? (float)(sums[d + c] * inv_count) : 0.f;
  }
  __syncthreads();
  const int CP = min(d / V, 256), RS = 256 / CP;
  const int ck = threadIdx.x % CP, rsub = threadIdx.x / CP;
  if (rsub >= RS) return;
  for (int c = ck * V; c < d; c += CP * V) {
    float mu[V], rs[V], g[V], bt[V], k1[V], k2[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      mu[j] = coef[c + j]; rs[j] = coef[d + c + j]; g[j] = coef[2 * d + c + j]; bt[j] = coef[3 * d + c + j];
      k1[j] = coef[4 * d + c + j]; k2[j] = coef[5 * d + c + j];
    }
#pragma unroll UNR
    for (long long m = (long long)blockIdx.x * RS + rsub; m < M; m += (long long)gridDim.x * RS) {
      float v[V], e[V], o[V];
      VecIO<TT>::load(x + m * d + c, v);
      VecIO<TT>::load(dy + m * d + c, e);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float xh = (v[j] - mu[j]) * rs[j];
        float dz = e[j] * swish_grad(g[j] * xh + bt[j]);
        dz -= k1[j] + xh * k2[j];
        o[j] = g[j] * rs[j] * dz;
      }
      VecIO<TT>::store(dx + m * d + c, o);
    }
  }
}
__global__ void bn_param_grad_kernel(const double* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta, int d) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d) return;
  dbeta[c] += (float)sums[c];
  dgamma[c] += (float)sums[d + c];
}

// =================================================================================================
// row-strided elementwise BN kernels: ~2 rows per thread
static inline int bn_grid(long long M, int d, int dt) {
  const int V = dt == MI_DT_BF16 ? 8 : 4;
  const int CP = d / V < 256 ? d / V : 256, RS = 256 / CP;
  long long g = (M + (long long)RS * 2 - 1) / ((long long)RS * 2);
  return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}
static int env_int_cm(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && e[0]) ? atoi(e) : dflt;
}
static inline int grid_for(long long n) { long long g = (n + 255) / 256; return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }

// MI355X_DWCONV_STREAM: 0 (default) = the LDS-tile kernels for every shape, 1 = the streaming kernel in the FORWARD pass (bf16,
// k = 31, even d), 2 = in the backward pass as well.  ALONE (warm operands) the streaming kernels win both ways (forward 26.5 ->
// 21 us, backward 46.5 -> 43 us per layer); INSIDE the training step both lose: the forward kernel reads its 31-tap window
// straight from global memory and counts on the cache for the 30 re-reads, which the step does not give it (43 us per launch in
// the step's kernel trace against 32.6 us of the tile kernel: profiles/r4_kernel_stats_per_step.md; same-box alternating runs of
// bench.py: 41.02 / 41.21 ms with 0, 41.92 / 41.90 ms with 1, profiles/r4_dwconv_in_step.md), and the backward pair loses
// 1.7 ms per step (42.46 vs 44.1 ms): its 512-thread / 207-register workgroups need a whole CU while half the CUs hold a
// workgroup of the weight-gradient stream's grouped GEMM for ~270 us at a time.  The micro-benchmark was the wrong judge twice.
static int g_dw_level = -1;  // -1: not yet read from the environment (set once per process, or by mi355x_dwconv_config from tests)
static int dw_stream_level() {
  if (g_dw_level < 0) {
    const char* e = getenv("MI355X_DWCONV_STREAM");
    g_dw_level = (e && e[0]) ? atoi(e) : 0;
  }
  return g_dw_level;
}
extern "C" int mi355x_dwconv_config(int level) {
  const int old = dw_stream_level();
  if (level >= 0) g_dw_level = level;
  return old;
}
static bool dw_stream_ok(int dt, int d, int ksize, int level) {
  return dw_stream_level() >= level && dt == MI_DT_BF16 && ksize == 31 && (d % 2) == 0;
}
extern "C" int mi355x_dwconv_fwd_ctx(const void* x, const void* w, const void* bias, void* y, int dt, void* stats, int B, int T,
                                     int d, int ksize, int pad_left, void* stream);
extern "C" int mi355x_dwconv_fwd(const void* x, const void* w, const void* bias, void* y, int dt, void* stats, int B, int T,
                                 int d, int ksize, void* stream) {
  return mi355x_dwconv_fwd_ctx(x, w, bias, y, dt, stats, B, T, d, ksize, -1, stream);
}
// pad_left: frames of zero padding in front of the sequence (conv_context_size = [left, right], left + right + 1 = ksize;
// causal: ksize - 1; parts/submodules/causal_convs.py:89-150); -1 = symmetric, (ksize - 1) / 2
extern "C" int mi355x_dwconv_fwd_ctx(const void* x, const void* w, const void* bias, void* y, int dt, void* stats, int B, int T,
                                     int d, int ksize, int pad_left, void* stream) {
  mi_clear_errors();
  if (!x || !w || !y || B <= 0 || T <= 0 || d <= 0 || pad_left < -1 || pad_left >= ksize) return MI_ERR_ARG;
  const int pad_shift = pad_left < 0 ? 0 : pad_left - (ksize - 1) / 2;
  dim3 grid((d + DW_CH - 1) / DW_CH, (T + DW_TT - 1) / DW_TT, B), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (pad_shift == 0 && dw_stream_ok(dt, d, ksize, 1)) {
    dim3 gs((d + DS_CG - 1) / DS_CG, (T + DS_BT - 1) / DS_BT, B), bs(64 * DS_NW);
#define DS_FWD(ST, DD) MI_LAUNCH((dwconv_stream_kernel<31, false, ST, DD>), gs, bs, 0, s, (const bf16_t*)x, (const float*)w, \
    (const float*)bias, (bf16_t*)y, (double*)stats, T, d)
    if (stats) { if (d == 512) DS_FWD(true, 512); else if (d == 256) DS_FWD(true, 256); else DS_FWD(true, 0); }
    else { if (d == 512) DS_FWD(false, 512); else if (d == 256) DS_FWD(false, 256); else DS_FWD(false, 0); }
#undef DS_FWD
    return mi_check_launch();
  }
  const DwGluArgs noglu = {nullptr, nullptr, nullptr, nullptr, 0, pad_shift};
#define DW_FWD(KS) DISPATCH_DT(dt, TT, MI_LAUNCH((dwconv_fwd_kernel<TT, KS, false>), grid, block, 0, s, (const TT*)x, \
    (const float*)w, (const float*)bias, (TT*)y, (double*)stats, B, T, d, noglu))
  switch (ksize) {
    case 31: DW_FWD(31); break;
    case 9: DW_FWD(9); break;
    case 5: DW_FWD(5); break;
    case 3: DW_FWD(3); break;
    default: return MI_ERR_ARG;
  }
#undef DW_FWD
  return mi_check_launch();
}
// GLU (+ pad mask) fused into the depthwise forward (conformer_modules.py:324-335: glu -> masked_fill -> depthwise_conv): glu_in
// [rows, 2d] is the pointwise conv's output (packed rows when row_offsets is given, as in mi355x_glu_fwd), glu_out [B,T,d] receives
// the GLU output (backward's operand), y / stats as in mi355x_dwconv_fwd.  One launch, and the GLU output is not read back.
extern "C" int mi355x_dwconv_fwd_glu(const void* glu_in, const void* len, const void* row_offsets, void* glu_out, const void* w,
                                     const void* bias, void* y, int dt, void* stats, int B, int T, int d, int ksize, int act,
                                     void* stream) {
  mi_clear_errors();
  if (!glu_in || !glu_out || !w || !y || B <= 0 || T <= 0 || d <= 0 || (d % (dt == MI_DT_BF16 ? 8 : 4)) || (row_offsets && !len) ||
      (act != 0 && act != 1))
    return MI_ERR_ARG;
  dim3 grid((d + DW_CH - 1) / DW_CH, (T + DW_TT - 1) / DW_TT, B), block(256);
  hipStream_t s = (hipStream_t)stream;
  const DwGluArgs glu = {glu_in, glu_out, (const long long*)len, (const long long*)row_offsets, act, 0};
#define DW_FWD_GLU(KS) DISPATCH_DT(dt, TT, MI_LAUNCH((dwconv_fwd_kernel<TT, KS, true>), grid, block, 0, s, (const TT*)nullptr, \
    (const float*)w, (const float*)bias, (TT*)y, (double*)stats, B, T, d, glu))
  switch (ksize) {
    case 31: DW_FWD_GLU(31); break;
    case 9: DW_FWD_GLU(9); break;
    case 5: DW_FWD_GLU(5); break;
    case 3: DW_FWD_GLU(3); break;
    default: return MI_ERR_ARG;
  }
#undef DW_FWD_GLU
  return mi_check_launch();
}
extern "C" int mi355x_dwconv_bwd_ctx(const void* dy, const void* x, const void* w, void* dx, void* dw, void* dbias, int dt, int B,
                                     int T, int d, int ksize, int pad_left, void* scratch, long long scratch_elems, void* stream);
extern "C" int mi355x_dwconv_bwd(const void* dy, const void* x, const void* w, void* dx, void* dw, void* dbias, int dt, int B,
                                 int T, int d, int ksize, void* scratch, long long scratch_elems, void* stream) {
  return mi355x_dwconv_bwd_ctx(dy, x, w, dx, dw, dbias, dt, B, T, d, ksize, -1, scratch, scratch_elems, stream);
}
extern "C" int mi355x_dwconv_bwd_ctx(const void* dy, const void* x, const void* w, void* dx, void* dw, void* dbias, int dt, int B,
                                     int T, int d, int ksize, int pad_left, void* scratch, long long scratch_elems, void* stream) {
  mi_clear_errors();
  if (!dy || !x || !w || !dx || !dw || B <= 0 || T <= 0 || d <= 0 || pad_left < -1 || pad_left >= ksize) return MI_ERR_ARG;
  const int pad_shift = pad_left < 0 ? 0 : pad_left - (ksize - 1) / 2;
  dim3 grid((d + DW_CH - 1) / DW_CH, DW_SEG, B), block(256);
  hipStream_t s = (hipStream_t)stream;
  const int nparts = B * DW_SEG;
  if (scratch && scratch_elems < (long long)nparts * (ksize + 1) * d) return MI_ERR_ARG;
  if (scratch && pad_shift == 0 && dw_stream_ok(dt, d, ksize, 2)) {
    // dx = the forward kernel with the taps flipped; the tap / bias gradient partials from their own streaming kernel
    dim3 gs((d + DS_CG - 1) / DS_CG, (T + DS_BT - 1) / DS_BT, B), bs(64 * DS_NW);
#define DS_DX(DD) MI_LAUNCH((dwconv_stream_kernel<31, true, false, DD>), gs, bs, 0, s, (const bf16_t*)dy, (const float*)w, \
    (const float*)nullptr, (bf16_t*)dx, (double*)nullptr, T, d)
    if (d == 512) DS_DX(512); else if (d == 256) DS_DX(256); else DS_DX(0);
#undef DS_DX
    const size_t shm = 4 * 32 * DS_CG * sizeof(float);
    static const bool attr_ok =
        hipFuncSetAttribute((const void*)dwconv_dw_stream_kernel<31, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) == hipSuccess &&
        hipFuncSetAttribute((const void*)dwconv_dw_stream_kernel<31, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) == hipSuccess &&
        hipFuncSetAttribute((const void*)dwconv_dw_stream_kernel<31, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) == hipSuccess;
    if (!attr_ok) return MI_ERR_LAUNCH;
    dim3 gw_((d + DS_CG - 1) / DS_CG, DW_SEG, B);
#define DS_DW(DD) MI_LAUNCH((dwconv_dw_stream_kernel<31, DD>), gw_, block, shm, s, (const bf16_t*)dy, (const bf16_t*)x, \
    (float*)scratch, T, d)
    if (d == 512) DS_DW(512); else if (d == 256) DS_DW(256); else DS_DW(0);
#undef DS_DW
    MI_LAUNCH(tap_reduce_kernel, dim3(((ksize + 1) * d + 255) / 256, 4), dim3(256), 0, s, (const float*)scratch, nparts,
                       ksize, d, (float*)dw, (float*)dbias);
    return mi_check_launch();
  }
  const DwBnArgs nobn = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, pad_shift};
#define DW_BWD(KS) do { if (pad_shift) { DISPATCH_DT(dt, TT, MI_LAUNCH((dwconv_bwd_kernel<TT, KS, false, true>), grid, block, 0, s, (const TT*)dy, \
    (const TT*)x, (const float*)w, (TT*)dx, (float*)dw, (float*)dbias, (float*)scratch, B, T, d, nobn)); } \
  else { DISPATCH_DT(dt, TT, MI_LAUNCH((dwconv_bwd_kernel<TT, KS, false, false>), grid, block, 0, s, (const TT*)dy, \
    (const TT*)x, (const float*)w, (TT*)dx, (float*)dw, (float*)dbias, (float*)scratch, B, T, d, nobn)); } } while (0)
  switch (ksize) {
    case 31: DW_BWD(31); break;
    case 9: DW_BWD(9); break;
    case 5: DW_BWD(5); break;
    case 3: DW_BWD(3); break;
    default: return MI_ERR_ARG;
  }
  if (scratch)
    MI_LAUNCH(tap_reduce_kernel, dim3(((ksize + 1) * d + 255) / 256, 4), dim3(256), 0, s, (const float*)scratch, nparts,
                       ksize, d, (float*)dw, (float*)dbias);
  return mi_check_launch();
}
// BatchNorm + Swish backward fused into the depthwise backward (conformer_modules.py:333-342 backward: swish -> batch_norm ->
// depthwise_conv): dy is the gradient w.r.t. the Swish output, cc the BatchNorm input, sums what mi355x_bn_swish_bwd_reduce left.
// Same results as mi355x_bn_swish_bwd_apply followed by mi355x_dwconv_bwd (the intermediate is rounded the same way), one launch and
// 2 x [B, T, d] of HBM traffic less.  count > 0, or count_dev (device f64) for SyncBatchNorm over ragged ranks.
extern "C" int mi355x_dwconv_bwd_bnswish(const void* dy, const void* cc, const void* mean, const void* rstd, const void* gamma,
                                         const void* beta, const void* sums, double count, const void* count_dev, int training,
                                         const void* x, const void* w, void* dx, void* dw, void* dbias, const void* glu_in,
                                         void* glu_din, const void* glu_len, const void* glu_row_offsets, int glu_act, int dt, int B,
                                         int T, int d, int ksize, void* scratch, long long scratch_elems, int defer_tap_reduce,
                                         void* stream) {
  mi_clear_errors();
  if (!dy || !cc || !mean || !rstd || !gamma || !beta || !sums || !x || !w || !dw || B <= 0 || T <= 0 || d <= 0) return MI_ERR_ARG;
  if (!count_dev && count <= 0) return MI_ERR_ARG;
  // either dx, or the GLU backward's output (glu_in + glu_din; then d must be a whole number of 16-byte chunks)
  if (glu_in ? (!glu_din || (d % (dt == MI_DT_BF16 ? 8 : 4)) || (glu_row_offsets && !glu_len) || (glu_act != 0 && glu_act != 1)) : !dx)
    return MI_ERR_ARG;
  dim3 grid((d + DW_CH - 1) / DW_CH, DW_SEG, B), block(256);
  hipStream_t s = (hipStream_t)stream;
  const int nparts = B * DW_SEG;
  if (scratch && scratch_elems < (long long)nparts * (ksize + 1) * d) return MI_ERR_ARG;
  const DwBnArgs bn = {cc, (const float*)mean, (const float*)rstd, (const float*)gamma, (const float*)beta, (const double*)sums,
                       count_dev ? 0.0 : 1.0 / count, (const double*)count_dev, training, glu_in, glu_din,
                       (const long long*)glu_len, (const long long*)glu_row_offsets, glu_act, 0};
#define DW_BWD_BN(KS) DISPATCH_DT(dt, TT, MI_LAUNCH((dwconv_bwd_kernel<TT, KS, true>), grid, block, 0, s, (const TT*)dy, \
    (const TT*)x, (const float*)w, (TT*)dx, (float*)dw, (float*)dbias, (float*)scratch, B, T, d, bn))
  switch (ksize) {
    case 31: DW_BWD_BN(31); break;
    case 9: DW_BWD_BN(9); break;
    case 5: DW_BWD_BN(5); break;
    case 3: DW_BWD_BN(3); break;
    default: return MI_ERR_ARG;
  }
#undef DW_BWD_BN
  if (defer_tap_reduce && !scratch) return MI_ERR_ARG;
  if (scratch && !defer_tap_reduce)   // (deferred: the caller runs mi355x_dwconv_tap_reduce on the stream of its choice)
    MI_LAUNCH(tap_reduce_kernel, dim3(((ksize + 1) * d + 255) / 256, 4), dim3(256), 0, s, (const float*)scratch, nparts,
                       ksize, d, (float*)dw, (float*)dbias);
  return mi_check_launch();
}
// second stage of the depthwise weight / bias gradient on its own: dw[c, k] += sum over the B * 4 partial slabs the backward kernels
// left in `scratch`.  Nothing on the backward chain reads dw / dbias -- only the optimizer does -- so a caller that gives every
// layer its own scratch can run this beside the chain (the weight-gradient stream) instead of inside it.
extern "C" int mi355x_dwconv_tap_reduce(const void* scratch, long long scratch_elems, int B, int d, int ksize, void* dw, void* dbias,
                                        void* stream) {
  mi_clear_errors();
  const int nparts = B * DW_SEG;
  if (!scratch || !dw || B <= 0 || d <= 0 || ksize <= 0 || scratch_elems < (long long)nparts * (ksize + 1) * d) return MI_ERR_ARG;
  MI_LAUNCH(tap_reduce_kernel, dim3(((ksize + 1) * d + 255) / 256, 4), dim3(256), 0, (hipStream_t)stream, (const float*)scratch,
            nparts, ksize, d, (float*)dw, (float*)dbias);
  return mi_check_launch();
}
static int bn_finalize_launch(const void* stats, double count, const void* count_dev, void* mean, void* rstd,
                              void* running_mean, void* running_var, float momentum, float eps, int d, void* stream) {
  mi_clear_errors();
  if (!stats || !mean || !rstd || d <= 0 || (!count_dev && count <= 0)) return MI_ERR_ARG;
  MI_LAUNCH(bn_finalize_kernel, dim3((d + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const double*)stats, count,
                     (const double*)count_dev, (float*)mean, (float*)rstd, (float*)running_mean, (float*)running_var, momentum,
                     eps, d);
  return mi_check_launch();
}
extern "C" int mi355x_bn_finalize(const void* stats, double count, void* mean, void* rstd, void* running_mean,
                                  void* running_var, float momentum, float eps, int d, void* stream) {
  return bn_finalize_launch(stats, count, nullptr, mean, rstd, running_mean, running_var, momentum, eps, d, stream);
}
extern "C" int mi355x_bn_finalize_dev_count(const void* stats, const void* count_dev, void* mean, void* rstd,
                                            void* running_mean, void* running_var, float momentum, float eps, int d,
                                            void* stream) {
  if (!count_dev) return MI_ERR_ARG;
  return bn_finalize_launch(stats, 0.0, count_dev, mean, rstd, running_mean, running_var, momentum, eps, d, stream);
}
extern "C" int mi355x_bn_eval_stats(const void* running_mean, const void* running_var, void* mean, void* rstd, float eps, int d,
                                    void* stream) {
  mi_clear_errors();
  if (!running_mean || !running_var || !mean || !rstd || d <= 0) return MI_ERR_ARG;
  MI_LAUNCH(bn_eval_stats_kernel, dim3((d + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)running_mean,
                     (const float*)running_var, (float*)mean, (float*)rstd, eps, d);
  return mi_check_launch();
}
extern "C" int mi355x_bn_swish_fwd(const void* x, const void* mean, const void* rstd, const void* gamma, const void* beta, void* y,
                                   int dt, long long M, int d, void* stream) {
  mi_clear_errors();
  if (!x || !mean || !rstd || !gamma || !beta || !y || M <= 0 || d <= 0 || d % (dt == MI_DT_BF16 ? 8 : 4)) return MI_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_DT(dt, TT, MI_LAUNCH((bn_swish_fwd_kernel<TT>), dim3(bn_grid(M, d, dt)), dim3(256), 0, s, (const TT*)x,
                                         (const float*)mean, (const float*)rstd, (const float*)gamma, (const float*)beta, (TT*)y, M, d));
  return mi_check_launch();
}
extern "C" int mi355x_bn_stats_swish_fwd(const void* x, const void* stats, double count, const void* count_dev, const void* gamma,
                                         const void* beta, void* y, void* mean, void* rstd, void* running_mean, void* running_var,
                                         float momentum, float eps, int dt, long long M, int d, void* stream) {
  mi_clear_errors();
  if (!x || !stats || !mean || !rstd || !gamma || !beta || !y || M <= 0 || d <= 0 || d % (dt == MI_DT_BF16 ? 8 : 4) ||
      (!count_dev && count <= 0))
    return MI_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_DT(dt, TT, MI_LAUNCH((bn_stats_swish_fwd_kernel<TT>), dim3(bn_grid(M, d, dt)), dim3(256), 0, s, (const TT*)x,
                                (const double*)stats, count, count_dev ? 0.0 : 1.0 / count, (const double*)count_dev, (const float*)gamma,
                                (const float*)beta, (TT*)y, (float*)mean, (float*)rstd, (float*)running_mean, (float*)running_var,
                                momentum, eps, M, d));
  return mi_check_launch();
}
// second stage of the BatchNorm-backward reduction: sums[i] (f64) += the slab column sums, and the parameter gradients (which
// are exactly these LOCAL sums: dbeta = sums[0:d], dgamma = sums[d:2d]) in the same pass -- no separate bn_param_grad launch
__global__ __launch_bounds__(256) void bn_partials_reduce_kernel(const float* __restrict__ partial, int nparts, int d,
                                                                 double* __restrict__ sums, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = 2 * d;
  if (i >= n) return;
  const int per = (nparts + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * per, p1 = min(nparts, p0 + per);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int p = p0;
  for (; p + 4 <= p1; p += 4) {
    a0 += partial[(long long)(p + 0) * n + i]; a1 += partial[(long long)(p + 1) * n + i];
    a2 += partial[(long long)(p + 2) * n + i]; a3 += partial[(long long)(p + 3) * n + i];
  }
  for (; p < p1; ++p) a0 += partial[(long long)p * n + i];
  const float acc = (a0 + a1) + (a2 + a3);
  atomicAdd(sums + i, (double)acc);
  if (dbeta) atomicAdd(i < d ? dbeta + i : dgamma + (i - d), acc);
}
extern "C" int mi355x_bn_swish_bwd_reduce(const void* dy, const void* x, const void* mean, const void* rstd, const void* gamma,
                                          const void* beta, void* sums, void* dgamma, void* dbeta, int dt, long long M, int d,
                                          void* scratch, long long scratch_elems, void* stream) {
  mi_clear_errors();
  if (!dy || !x || !sums || M <= 0 || d <= 0 || (!dgamma != !dbeta)) return MI_ERR_ARG;
  const int V = dt == MI_DT_BF16 ? 8 : 4;
  if (d % V) return MI_ERR_ARG;
  // rows per workgroup: 32 (501 workgroups at the Large shape) when the scratch slab is the documented ceil(M/32)*2*d; a larger slab
  // lets MI355X_BNR_ROWS = 16 double the workgroups (tools/bn_bench.py)
  static const int env_rows = env_int_cm("MI355X_BNR_ROWS", BNR_ROWS);
  int rows = env_rows >= 8 && env_rows <= 256 ? env_rows : BNR_ROWS;
  if (scratch && scratch_elems < (long long)((M + rows - 1) / rows) * 2 * d) rows = BNR_ROWS;
  const unsigned nblk = (unsigned)((M + rows - 1) / rows);
  if (scratch && scratch_elems < (long long)nblk * 2 * d) return MI_ERR_ARG;
  dim3 grid(1, nblk), block(256);
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_DT(dt, TT, MI_LAUNCH((bn_swish_bwd_reduce_kernel<TT>), grid, block, 0, s, (const TT*)dy, (const TT*)x,
                                         (const float*)mean, (const float*)rstd, (const float*)gamma, (const float*)beta,
                                         (double*)sums, (float*)scratch, M, d, rows));
  if (scratch)
    MI_LAUNCH(bn_partials_reduce_kernel, dim3((2 * d + 255) / 256, 32), dim3(256), 0, s, (const float*)scratch, (int)nblk, d,
              (double*)sums, (float*)dgamma, (float*)dbeta);
  else if (dgamma)  // (single-stage path: the sums are complete only after the kernel above)
    MI_LAUNCH(bn_param_grad_kernel, dim3((d + 255) / 256), dim3(256), 0, s, (const double*)sums, (float*)dgamma, (float*)dbeta, d);
  return mi_check_launch();
}
static int bn_swish_bwd_apply_launch(const void* dy, const void* x, const void* mean, const void* rstd, const void* gamma,
                                     const void* beta, const void* sums, double count, const void* count_dev, int training,
                                     void* dx, int dt, long long M, int d, void* stream);
extern "C" int mi355x_bn_swish_bwd_apply(const void* dy, const void* x, const void* mean, const void* rstd, const void* gamma,
                                         const void* beta, const void* sums, double count, int training, void* dx, int dt,
                                         long long M, int d, void* stream) {
  return bn_swish_bwd_apply_launch(dy, x, mean, rstd, gamma, beta, sums, count, nullptr, training, dx, dt, M, d, stream);
}
extern "C" int mi355x_bn_swish_bwd_apply_dev_count(const void* dy, const void* x, const void* mean, const void* rstd,
                                                   const void* gamma, const void* beta, const void* sums, const void* count_dev,
                                                   int training, void* dx, int dt, long long M, int d, void* stream) {
  if (!count_dev) return MI_ERR_ARG;
  return bn_swish_bwd_apply_launch(dy, x, mean, rstd, gamma, beta, sums, 0.0, count_dev, training, dx, dt, M, d, stream);
}
static int bn_swish_bwd_apply_launch(const void* dy, const void* x, const void* mean, const void* rstd, const void* gamma,
                                     const void* beta, const void* sums, double count, const void* count_dev, int training,
                                     void* dx, int dt, long long M, int d, void* stream) {
  mi_clear_errors();
  if (!dy || !x || !sums || !dx || M <= 0 || d <= 0 || d % (dt == MI_DT_BF16 ? 8 : 4) || (!count_dev && count <= 0))
    return MI_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if ((size_t)d * 6 * sizeof(float) > 64 * 1024) return MI_ERR_ARG;
  // rows per thread (MI355X_BNA_ROWS, default 2) and the unroll of the row loop (MI355X_BNA_UNR: 2 | 4): every workgroup derives the
  // coefficients of all d channels before its first row, so fewer, longer workgroups amortise that set-up (tools/bn_bench.py)
  static const int rpt = env_int_cm("MI355X_BNA_ROWS", 2), unr = env_int_cm("MI355X_BNA_UNR", 2);
  const int V = dt == MI_DT_BF16 ? 8 : 4;
  const int CP = d / V < 256 ? d / V : 256, RS = 256 / CP;
  long long g = (M + (long long)RS * rpt - 1) / ((long long)RS * (rpt > 0 ? rpt : 2));
  g = g > 16384 ? 16384 : (g < 1 ? 1 : g);
#define BNA_LAUNCH(UNR_) DISPATCH_DT(dt, TT, MI_LAUNCH((bn_swish_bwd_apply_kernel<TT, UNR_>), dim3((unsigned)g), dim3(256), \
    (size_t)d * 6 * sizeof(float), s, (const TT*)dy, (const TT*)x, (const float*)mean, (const float*)rstd, (const float*)gamma, \
    (const float*)beta, (const double*)sums, count_dev ? 0.0 : 1.0 / count, (const double*)count_dev, training, (TT*)dx, M, d))
  if (unr == 4) { BNA_LAUNCH(4); } else { BNA_LAUNCH(2); }
#undef BNA_LAUNCH
  return mi_check_launch();
}
extern "C" int mi355x_bn_param_grad(const void* sums, void* dgamma, void* dbeta, int d, void* stream) {
  mi_clear_errors();
  if (!sums || !dgamma || !dbeta || d <= 0) return MI_ERR_ARG;
  MI_LAUNCH(bn_param_grad_kernel, dim3((d + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const double*)sums,
                     (float*)dgamma, (float*)dbeta, d);
  return mi_check_launch();
}
