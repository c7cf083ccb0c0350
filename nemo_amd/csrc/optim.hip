// Optimizer + weight-packing kernels (HBM-bound, flat buffers sized for 288 GB HBM: one launch per step each).
//   * fused AdamW over the flat fp32 parameter / gradient / moment buffers (decoupled weight decay, bias correction,
//     optional global grad scale = 1/world for the data-parallel average)  -- replaces torch.optim.AdamW stepping
//     646+ tensors (nemo/core/classes/modelPT.py:650-823, optim registry nemo/core/optim/optimizers.py:33)
//   * descriptor-driven "pack" kernel that produces every bf16 GEMM operand image of the weights in one launch:
//     plain casts, transposes (for dgrad), q|k|v concatenation, conv2 [co,ci,3,3] -> [co][(kh,kw,ci)], and the
//     out-Linear column permutation (c*F2+f -> f*C+c) required by the channels-last conv layout.
#include "common.h"
#include "mi355x_asr.h"

// gclip: optional device scalar multiplied into the gradient (global-norm clipping coefficient, no host sync);
// ema: optional exponential moving average of the weights, updated in the same pass (ema.py:150-157)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long long n4, float lr, float beta1, float beta2,
                                                    float eps, float wd, float bc1, float bc2_sqrt, float grad_scale,
                                                    const float* __restrict__ gclip, float* __restrict__ ema, float ema_decay) {
  const float gs = gclip ? grad_scale * gclip[0] : grad_scale;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float pp[4], gg[4], mm[4], vv[4];
    ld4(p + i * 4, pp); ld4(g + i * 4, gg); ld4(m + i * 4, mm); ld4(v + i * 4, vv);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gr = gg[j] * gs;
      pp[j] *= (1.f - lr * wd);
      mm[j] = beta1 * mm[j] + (1.f - beta1) * gr;
      vv[j] = beta2 * vv[j] + (1.f - beta2) * gr * gr;
      const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
      pp[j] -= (lr / bc1) * (mm[j] / denom);
    }
    st4(p + i * 4, pp); st4(m + i * 4, mm); st4(v + i * 4, vv);
    if (ema) {
      float ee[4];
      ld4(ema + i * 4, ee);
#pragma unroll
      for (int j = 0; j < 4; ++j) ee[j] = ee[j] * ema_decay + (1.f - ema_decay) * pp[j];
      st4(ema + i * 4, ee);
    }
  }
}
// sum of squares of a flat gradient buffer -> out[0] += (f64); block partials in f32, one f64 atomic per block
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n4, double* __restrict__ out) {
  __shared__ float red[8];
  float a = 0.f;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float x[4];
    ld4(g + i * 4, x);
    a += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
  }
  a = block_sum(a, red);
  if (threadIdx.x == 0) atomicAdd(out, (double)a);
}
// torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm / (scale * sqrt(sum of the buffers' sums) + 1e-6))
__global__ void clip_coef_kernel(const double* __restrict__ sumsq, int nbuf, float scale, float max_norm, float* __restrict__ coef) {
  double t = 0.0;
  for (int i = 0; i < nbuf; ++i) t += sumsq[i];
  const float norm = scale * (float)sqrt(t);
  const float c = max_norm / (norm + 1e-6f);
  coef[0] = c < 1.f ? c : 1.f;
  coef[1] = norm;
}

// dst[r*pitch + c] = cast( src[r1*sr1 + r2*sr2 + c1*sc1 + c2*sc2] ),  r = r1*nr2 + r2,  c = c1*nc2 + c2
struct PackEntry {
  const float* src; void* dst;
  int rows, cols, nr2, nc2;
  long long sr1, sr2, sc1, sc2, pitch;
  long long tile_begin;  // prefix sum of 32x32 tiles
};
#define PK_T 64  // tile edge; tile_begin counts 64x64 tiles
__global__ __launch_bounds__(256) void pack_kernel(const PackEntry* __restrict__ tab, int n_entries, int out_dt) {
  // 64x64 tiles through LDS: reads run along whichever axis is contiguous in the fp32 source (256-B wave segments),
  // writes are 8 consecutive image columns per lane (16-B bf16 / 32-B f32 stores; 2-byte stores crawl on this chip)
  __shared__ float tile[PK_T][PK_T + 1];
  int lo = 0, hi = n_entries - 1;
  const long long bid = blockIdx.x;
  while (lo < hi) {  // locate the entry by binary search on tile_begin
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].tile_begin <= bid) lo = mid; else hi = mid - 1;
  }
  const PackEntry e = tab[lo];
  const int tiles_c = (e.cols + PK_T - 1) / PK_T;
  const int local = (int)(bid - e.tile_begin);
  const int tr = local / tiles_c, tc = local - tr * tiles_c;
  const bool col_fast = (e.sc2 == 1 && e.nc2 > 1) || (e.nc2 == 1 && e.sc1 == 1);
  const bool plain = e.nr2 == 1 && e.nc2 == 1;  // (uniform) a strided 2-D view: no index splitting, and ...
  const long long fast_stride = col_fast ? e.sc1 : e.sr1, slow_stride = col_fast ? e.sr1 : e.sc1;
  const int fast_n = col_fast ? e.cols : e.rows, slow_n = col_fast ? e.rows : e.cols;
  const int f0 = (col_fast ? tc : tr) * PK_T, s0 = (col_fast ? tr : tc) * PK_T;
  if (plain && fast_stride == 1 && !(slow_stride & 3) && !((uintptr_t)e.src & 15) && f0 + PK_T <= fast_n) {
    // ... whole 16-byte pieces along the contiguous axis of the source: 4 float4 loads per thread instead of 16 scalar ones
    // with two integer divisions each (the divisions, not the memory system, bounded this kernel: 2.4 TB/s)
#pragma unroll
    for (int k = 0; k < PK_T * PK_T / 4 / 256; ++k) {
      const int q = threadIdx.x + k * 256;
      const int f4 = (q & 15) * 4, sl = q >> 4;  // 16 lanes x 16 B = one 256-B row segment of the tile
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (s0 + sl < slow_n) v = *reinterpret_cast<const float4*>(e.src + (long long)(s0 + sl) * slow_stride + f0 + f4);
      if (col_fast) { tile[sl][f4] = v.x; tile[sl][f4 + 1] = v.y; tile[sl][f4 + 2] = v.z; tile[sl][f4 + 3] = v.w; }
      else { tile[f4][sl] = v.x; tile[f4 + 1][sl] = v.y; tile[f4 + 2][sl] = v.z; tile[f4 + 3][sl] = v.w; }
    }
  } else {
#pragma unroll 4
    for (int k = 0; k < PK_T * PK_T / 256; ++k) {
      const int q = threadIdx.x + k * 256;
      const int rr = col_fast ? (q >> 6) : (q & 63), cc = col_fast ? (q & 63) : (q >> 6);
      const int r = tr * PK_T + rr, c = tc * PK_T + cc;
      float v = 0.f;
      if (r < e.rows && c < e.cols) {
        if (plain) v = e.src[r * e.sr1 + c * e.sc1];
        else {
          const int r1 = r / e.nr2, r2 = r - r1 * e.nr2, c1 = c / e.nc2, c2 = c - c1 * e.nc2;
          v = e.src[r1 * e.sr1 + r2 * e.sr2 + c1 * e.sc1 + c2 * e.sc2];
        }
      }
      tile[rr][cc] = v;
    }
  }
  __syncthreads();
  // image pitch is a multiple of 8 and >= roundup8(cols): whole 8-column groups can be written (zeros past `cols`)
#pragma unroll
  for (int k = 0; k < PK_T * PK_T / 8 / 256; ++k) {
    const int q = threadIdx.x + k * 256;
    const int rr = q >> 3, c8 = (q & 7) * 8;
    const int r = tr * PK_T + rr, c = tc * PK_T + c8;
    if (r < e.rows && c < e.cols) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tile[rr][c8 + j];
      if (out_dt == MI_DT_F32) {
        float* d = (float*)e.dst + (long long)r * e.pitch + c;
        *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(d + 4) = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        u32x4 t = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
        *reinterpret_cast<u32x4*>((bf16_t*)e.dst + (long long)r * e.pitch + c) = t;
      }
    }
  }
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ p, long long n, float value) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = value;
}

static int adamw_launch(void* params, const void* grads, void* exp_avg, void* exp_avg_sq, long long n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int step, float grad_scale, const void* gclip, void* ema,
                        float ema_decay, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || n <= 0 || (n & 3) || step < 1) return MI_ERR_ARG;
  if (ema && !(ema_decay >= 0.f && ema_decay <= 1.f)) return MI_ERR_ARG;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  long long nb = ((n >> 2) + 255) / 256;
  if (nb > 8192) nb = 8192;
  MI_LAUNCH(adamw_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (float*)params, (const float*)grads,
                     (float*)exp_avg, (float*)exp_avg_sq, n >> 2, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale,
                     (const float*)gclip, (float*)ema, ema_decay);
  return mi_check_launch();
}
extern "C" int mi355x_adamw_step(void* params, const void* grads, void* exp_avg, void* exp_avg_sq, long long n, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                                 void* stream) {
  mi_clear_errors();
  return adamw_launch(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, nullptr,
                      nullptr, 0.f, stream);
}
extern "C" int mi355x_adamw_step_ex(void* params, const void* grads, void* exp_avg, void* exp_avg_sq, long long n, float lr,
                                    float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                                    const void* clip_coef, void* ema, float ema_decay, void* stream) {
  mi_clear_errors();
  return adamw_launch(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, clip_coef,
                      ema, ema_decay, stream);
}
extern "C" int mi355x_grad_sumsq(const void* grads, long long n, void* out_f64, void* stream) {
  mi_clear_errors();
  if (!grads || !out_f64 || n <= 0 || (n & 3)) return MI_ERR_ARG;
  long long nb = ((n >> 2) + 255) / 256;
  if (nb > 2048) nb = 2048;
  MI_LAUNCH(sumsq_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const float*)grads, n >> 2,
                     (double*)out_f64);
  return mi_check_launch();
}
extern "C" int mi355x_clip_coef(const void* sumsq_f64, int nbuf, float scale, float max_norm, void* coef_f32x2, void* stream) {
  mi_clear_errors();
  if (!sumsq_f64 || !coef_f32x2 || nbuf <= 0 || !(max_norm > 0.f)) return MI_ERR_ARG;
  MI_LAUNCH(clip_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const double*)sumsq_f64, nbuf, scale, max_norm,
                     (float*)coef_f32x2);
  return mi_check_launch();
}

extern "C" int mi355x_pack_weights(const void* table_dev, int n_entries, long long total_tiles, int out_dtype, void* stream) {
  mi_clear_errors();
  if (!table_dev || n_entries <= 0 || total_tiles <= 0) return MI_ERR_ARG;
  MI_LAUNCH(pack_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, (const PackEntry*)table_dev,
                     n_entries, out_dtype);
  return mi_check_launch();
}

extern "C" int mi355x_fill_f32(void* p, long long n, float value, void* stream) {
  mi_clear_errors();
  if (!p || n <= 0) return MI_ERR_ARG;
  long long nb = (n + 255) / 256;
  if (nb > 8192) nb = 8192;
  MI_LAUNCH(fill_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (float*)p, n, value);
  return mi_check_launch();
}
