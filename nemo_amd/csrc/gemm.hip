// GEMM kernels for the Conformer-CTC hot path on MI355X (gfx950).
//
//   C[M,N] = epilogue( sum_k opA(m,k) * opB(n,k) )          (optionally batched / split-K / gathered / grouped)
//
// bf16 (MFMA `v_mfma_f32_32x32x16_bf16`), four structures behind one descriptor (`mi355x_gemm`):
//   * gemm_bf16_v2_kernel   256x128x64 tile, 8 waves x (64x64), three LDS stages filled by LDS-DMA two K-tiles ahead
//                           (counted vmcnt, raw s_barrier), ds_read_b128 / ds_read_b64_tr_b16 fragments  -- the default
//   * gemm_bf16_v4_kernel   256x256x64 tile, 8 waves x (128x64), two LDS stages -- when the larger tile still fills the chip
//   * gemm_bf16_grouped_tn_kernel   up to 12 weight-gradient problems in one launch of the v2 body (`mi355x_gemm_grouped`)
//   * gemm_bf16_kernel      128x128x64 tile, 4 waves, register-staged -- small problems and the TN / K-contiguous-B layout
// An operand may be stored "reduction-major" ([K][rows]: both wgrad operands); it is DMA'd as it lies in memory and
// transposed by the LDS read.  Operand A (forward / dgrad) or the reduction-major B (wgrad) of a convolution can be
// GATHERED from a channels-last grid by the LDS-DMA (implicit GEMM), and output rows can be scattered (row map).
// Tile -> workgroup mapping is XCD-aware (bijective chunking of the tile list over the 8 XCDs so that tiles sharing an
// operand panel hit the same L2).
// f32 path: exact-fp32 VALU tile kernel with arbitrary strides (parity / fp32 configuration).
// All share the epilogue kinds below (bias, Swish+dropout, residual, Swish-grad, ReLU+time-mask, ReLU-grad, atomic split-K).
//
// Replaces on the reference path: torch.nn.functional.linear / conv1d(k=1) / matmul / conv2d calls in
//   nemo/collections/asr/parts/submodules/conformer_modules.py:382-387 (FFN), :321,343 (pointwise convs),
//   multi_head_attention.py:124-146,300-350 (q/k/v/pos/out projections; QK^T, PV on the fp32 path),
//   subsampling.py:431 (out Linear), :231-253 (conv2), modules/conv_asr.py:445 (decoder).
#include <stdlib.h>
#include "common.h"
#include <atomic>
#include "mi355x_asr.h"

enum {
  EPI_STORE = 0,       // C = alpha*dropout(acc+bias)
  EPI_SWISH_DROP = 1,  // aux_out = acc+bias ; C = dropout(swish(acc+bias))
  EPI_RESID = 2,       // C(f32) = aux_in(f32) + alpha*dropout(acc+bias)
  EPI_DSWISH = 3,      // C = acc * dropmask * swish'(aux_in)
  EPI_RELU_MASK = 4,   // C = relu(acc+bias) * (t(m) < len[b(m)])
  EPI_MUL_POS = 5,     // C = acc * (aux_in > 0)
};

#define BM 128
#define BN 128
#define BK 64

struct GemmP {
  const void* A; const void* B; void* C;
  int M, N, K;
  long long lda, ldb, ldc, csc;       // csc: column stride of C (1 = dense rows)
  int transA, transB;                 // 1: operand stored [K][rows] (reduction-major)
  int batch, nb0;                     // z -> z0 = z % nb0, z1 = z / nb0
  long long sA0, sA1, sB0, sB1, sC0, sC1;
  const float* bias; float alpha;
  int epi; int c_dt; int atomic;
  const void* aux_in; int auxin_dt; void* aux_out; int auxout_dt; long long ldaux;
  DropCfg drop;
  int swish_g;                         // see swish_fwd8 / the EPI_DSWISH sites: aux = swish'(h) * mask instead of h
  const long long* row_len; int rows_per_b; int rows_inner;
  int splitk; int ktiles_per_split;
  long long colsum_stride;             // batch stride (z0) of colsum_out
  float* colsum_out;                   // transA only: colsum_out[m] += sum_k A(k, m)  (bias gradient fused into wgrad)
  int vec_ok;                          // C / aux rows are 8-element aligned & dense: vectorised epilogue allowed
  int v8_delay;                        // eighth structure: start delay of every other first-round workgroup (10-ns ticks), see there
  // implicit-GEMM convolution (channels-last): A row m = (b, i, j) on an [nI x nJ] grid gathers, for K index tap*C + c,
  // src[b][i*si + di[tap]][j*sj + dj[tap]][c] of a [SI x SJ x C] source grid (zero outside)
  int g_on, g_nI, g_nJ, g_SI, g_SJ, g_C, g_si, g_sj, g_ntaps;
  // tap offsets, 4 bits per tap biased by 8 (|d| <= 7, <= 9 taps): decoded with scalar shifts.  (A byte table indexed by the
  // K-tile's tap compiled to a VECTOR byte load of the kernel arguments + `s_waitcnt vmcnt(0)` inside the K loop -- which also
  // waited for every LDS-DMA in flight: the conv2 loops ran without any prefetch overlap.)
  unsigned long long g_dip, g_djp;
  // output row map: C / aux row of m = (b, i, j) is ((b*OI + i*si + oi)*OJ + j*sj + oj)
  int r_on, r_nI, r_nJ, r_OI, r_OJ, r_si, r_sj, r_oi, r_oj;
};

__device__ __forceinline__ int tap_delta(unsigned long long packed, int tap) {  // tap must be wave-uniform
  return (int)((packed >> (__builtin_amdgcn_readfirstlane(tap) * 4)) & 15ull) - 8;
}

// storage row (C / aux) of logical row m
// q = m / d for 0 <= m < 2^24, 1 <= d < 2^24 without the ~40-instruction integer division sequence: float estimate (exact to
// +-1 for these ranges) and one correction step.  The row map is evaluated per row and tensor in the epilogue of the four conv2
// input-gradient GEMMs: with plain `/` it was most of that epilogue's VALU work.
__device__ __forceinline__ int fdiv24(int m, int d, float inv_d) {
  int q = (int)((float)m * inv_d);
  const int r = m - q * d;
  q += (r >= d) - (r < 0);
  return q;
}
__device__ __forceinline__ long long crow(const GemmP& p, int m) {
  if (!p.r_on) return m;
  const int per_b = p.r_nI * p.r_nJ;
  int b, i;
  if ((unsigned)p.M < (1u << 24)) {  // (uniform)
    b = fdiv24(m, per_b, __builtin_amdgcn_rcpf((float)per_b));
    const int r = m - b * per_b;
    i = fdiv24(r, p.r_nJ, __builtin_amdgcn_rcpf((float)p.r_nJ));
  } else {
    b = m / per_b;
    i = (m - b * per_b) / p.r_nJ;
  }
  const int r = m - b * per_b;
  const int j = r - i * p.r_nJ;
  return ((long long)b * p.r_OI + i * p.r_si + p.r_oi) * p.r_OJ + j * p.r_sj + p.r_oj;
}

__device__ __forceinline__ float ldx(const void* p, long long i, int dt) {
  return dt == MI_DT_F32 ? ((const float*)p)[i] : bf2f(((const bf16_t*)p)[i]);
}
__device__ __forceinline__ void stx(void* p, long long i, int dt, float v) {
  if (dt == MI_DT_F32) ((float*)p)[i] = v; else ((bf16_t*)p)[i] = f2bf(v);
}

// ---- Swish epilogues of the feed-forward pair.  Default (swish_g = 0): the forward GEMM keeps the pre-activation h and the
// backward GEMM's epilogue evaluates swish'(h) and re-creates the dropout mask per element -- exp + rcp + the mask hash, ~23 of its
// ~26 vector-op slots per element, 10 us per 256 x 256 tile on top of a 16-us K loop (profiles/r3_gemm_structures.md section 4).
// swish_g = 1: the forward epilogue, which has sigmoid(h) and the mask in registers anyway, stores g = swish'(h) * mask (bf16, the
// same 2 bytes per element) and the backward epilogue is one multiply: dh = acc * g.  g is rounded to bf16 once more than
// swish'(bf16 h) would be -- the same class of rounding as every other stored activation of the bf16 path.
__device__ __forceinline__ void swish_pair(float h, float dm, float& act, float& g) {
  const float s = sigmoidf_(h);
  const float sw = h * s;
  act = sw * dm;
  g = fmaf(sw, 1.f - s, s) * dm;
}

__device__ __forceinline__ void epilogue(const GemmP& p, int z, long long coff, int m, int n, float acc) {
  const long long mr = crow(p, m);
  const long long ci = coff + mr * p.ldc + (long long)n * p.csc;
  const long long ai = coff + mr * p.ldaux + n;  // aux shares the batch offset convention of C
  float v = acc;
  if (p.bias) v += p.bias[n];
  const uint32_t didx = (uint32_t)z * (uint32_t)(p.M * p.N) + (uint32_t)m * (uint32_t)p.N + (uint32_t)n;
  switch (p.epi) {
    case EPI_STORE: v *= p.alpha * drop_mask(p.drop, didx); break;
    case EPI_SWISH_DROP:
      if (p.swish_g) {
        float a_, g_;
        swish_pair(v, drop_mask(p.drop, didx), a_, g_);
        stx(p.aux_out, ai, p.auxout_dt, g_);
        v = a_;
      } else {
        stx(p.aux_out, ai, p.auxout_dt, v);
        v = swishf_(v) * drop_mask(p.drop, didx);
      }
      break;
    case EPI_RESID:
      v = ((const float*)p.aux_in)[ai] + p.alpha * v * drop_mask(p.drop, didx);
      break;
    case EPI_DSWISH:
      v = p.swish_g ? v * ldx(p.aux_in, ai, p.auxin_dt) : v * drop_mask(p.drop, didx) * swish_grad(ldx(p.aux_in, ai, p.auxin_dt));
      break;
    case EPI_RELU_MASK: {
      int b = m / p.rows_per_b;
      int t = (m - b * p.rows_per_b) / p.rows_inner;
      v = (v > 0.f && (long long)t < p.row_len[b]) ? v : 0.f;
    } break;
    case EPI_MUL_POS: v = ldx(p.aux_in, ai, p.auxin_dt) > 0.f ? v : 0.f; break;
  }
  if (p.atomic) atomicAdd(&((float*)p.C)[ci], v);
  else stx(p.C, ci, p.c_dt, v);
}

// ---- 8-wide epilogue (one thread = 8 consecutive columns of one row; 16-B / 32-B global accesses)
__device__ __forceinline__ void ld8x(const void* p, long long i, int dt, float (&v)[8]) {
  if (dt == MI_DT_F32) {
    const float4 a = *reinterpret_cast<const float4*>((const float*)p + i);
    const float4 b = *reinterpret_cast<const float4*>((const float*)p + i + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
    const u32x4 t = *reinterpret_cast<const u32x4*>((const bf16_t*)p + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(t[j] << 16); v[2 * j + 1] = __uint_as_float(t[j] & 0xffff0000u); }
  }
}
__device__ __forceinline__ void st8x(void* p, long long i, int dt, const float (&v)[8]) {
  if (dt == MI_DT_F32) {
    *reinterpret_cast<float4*>((float*)p + i) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>((float*)p + i + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    u32x4 t = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
    *reinterpret_cast<u32x4*>((bf16_t*)p + i) = t;
  }
}
__device__ __forceinline__ void epilogue8(const GemmP& p, int z, long long coff, int m, int n, float (&v)[8]) {
  const long long mr = crow(p, m);
  const long long ci = coff + mr * p.ldc + n;
  const long long ai = coff + mr * p.ldaux + n;
  if (p.bias) {
    float b[8];
    ld8x(p.bias, n, MI_DT_F32, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += b[j];
  }
  const uint32_t didx = (uint32_t)z * (uint32_t)(p.M * p.N) + (uint32_t)m * (uint32_t)p.N + (uint32_t)n;
  float dm[8];
  drop_mask8u(p.drop, didx, dm);  // (a width of 4 modulo 8 starts every other row in the middle of a hash group)
  switch (p.epi) {
    case EPI_STORE:
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= p.alpha * dm[j];
      break;
    case EPI_SWISH_DROP:
      if (p.swish_g) {
        float g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) swish_pair(v[j], dm[j], v[j], g[j]);
        st8x(p.aux_out, ai, p.auxout_dt, g);
      } else {
        st8x(p.aux_out, ai, p.auxout_dt, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = swishf_(v[j]) * dm[j];
      }
      break;
    case EPI_RESID: {
      float r[8];
      ld8x(p.aux_in, ai, MI_DT_F32, r);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = r[j] + p.alpha * v[j] * dm[j];
    } break;
    case EPI_DSWISH: {
      float h[8];
      ld8x(p.aux_in, ai, p.auxin_dt, h);
      if (p.swish_g) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= h[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * dm[j] * swish_grad(h[j]);
      }
    } break;
    case EPI_RELU_MASK: {
      const int b = m / p.rows_per_b;
      const int t = (m - b * p.rows_per_b) / p.rows_inner;
      const bool ok = (long long)t < p.row_len[b];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (ok && v[j] > 0.f) ? v[j] : 0.f;
    } break;
    case EPI_MUL_POS: {
      float h[8];
      ld8x(p.aux_in, ai, p.auxin_dt, h);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = h[j] > 0.f ? v[j] : 0.f;
    } break;
  }
  if (p.atomic) {
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(&((float*)p.C)[ci + j], v[j]);
  } else {
    st8x(p.C, ci, p.c_dt, v);
  }
}


// ---- 4-wide form of the above for the last columns of a matrix whose width is 4 modulo 8 (16-byte f32 / 8-byte bf16 accesses;
// `vec_ok & 2`): d = 324 outputs paid four scalar epilogues -- each with its own 64-bit address arithmetic and full mask hash -- per
// row for them, and every tile of the launch waited for the column block that holds them
__device__ __forceinline__ void ld4x(const void* p, long long i, int dt, float (&v)[4]) {
  if (dt == MI_DT_F32) {
    const float4 a = *reinterpret_cast<const float4*>((const float*)p + i);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  } else {
    const u32x2 t = *reinterpret_cast<const u32x2*>((const bf16_t*)p + i);
    v[0] = __uint_as_float(t[0] << 16); v[1] = __uint_as_float(t[0] & 0xffff0000u);
    v[2] = __uint_as_float(t[1] << 16); v[3] = __uint_as_float(t[1] & 0xffff0000u);
  }
}
__device__ __forceinline__ void st4x(void* p, long long i, int dt, const float (&v)[4]) {
  if (dt == MI_DT_F32) *reinterpret_cast<float4*>((float*)p + i) = make_float4(v[0], v[1], v[2], v[3]);
  else {
    u32x2 t = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
    *reinterpret_cast<u32x2*>((bf16_t*)p + i) = t;
  }
}
__device__ __forceinline__ void epilogue4(const GemmP& p, int z, long long coff, int m, int n, float (&v)[4]) {
  const long long mr = crow(p, m);
  const long long ci = coff + mr * p.ldc + n;
  const long long ai = coff + mr * p.ldaux + n;
  if (p.bias) {
    float b[4];
    ld4x(p.bias, n, MI_DT_F32, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += b[j];
  }
  const uint32_t didx = (uint32_t)z * (uint32_t)(p.M * p.N) + (uint32_t)m * (uint32_t)p.N + (uint32_t)n;
  float dm[4];
  drop_mask4u(p.drop, didx, dm);
  switch (p.epi) {
    case EPI_STORE:
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] *= p.alpha * dm[j];
      break;
    case EPI_SWISH_DROP:
      if (p.swish_g) {
        float g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) swish_pair(v[j], dm[j], v[j], g[j]);
        st4x(p.aux_out, ai, p.auxout_dt, g);
      } else {
        st4x(p.aux_out, ai, p.auxout_dt, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = swishf_(v[j]) * dm[j];
      }
      break;
    case EPI_RESID: {
      float r[4];
      ld4x(p.aux_in, ai, MI_DT_F32, r);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = r[j] + p.alpha * v[j] * dm[j];
    } break;
    case EPI_DSWISH: {
      float h[4];
      ld4x(p.aux_in, ai, p.auxin_dt, h);
      if (p.swish_g) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= h[j];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] * dm[j] * swish_grad(h[j]);
      }
    } break;
    case EPI_RELU_MASK: {
      const int b = m / p.rows_per_b;
      const int t = (m - b * p.rows_per_b) / p.rows_inner;
      const bool ok = (long long)t < p.row_len[b];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (ok && v[j] > 0.f) ? v[j] : 0.f;
    } break;
    case EPI_MUL_POS: {
      float h[4];
      ld4x(p.aux_in, ai, p.auxin_dt, h);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = h[j] > 0.f ? v[j] : 0.f;
    } break;
  }
  if (p.atomic) {
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(&((float*)p.C)[ci + j], v[j]);
  } else {
    st4x(p.C, ci, p.c_dt, v);
  }
}
// one 8-column chunk of a row, however many of its columns exist: 8-wide, 4-wide + scalars, or scalars
__device__ __forceinline__ void epilogue_chunk(const GemmP& p, int z, long long coff, int m, int n, float (&v)[8]) {
  const int nv = p.N - n;
  if ((p.vec_ok & 1) && nv >= 8) { epilogue8(p, z, coff, m, n, v); return; }
  int j0 = 0;
  if ((p.vec_ok & 2) && nv >= 4) {
    float w[4] = {v[0], v[1], v[2], v[3]};
    epilogue4(p, z, coff, m, n, w);
    j0 = 4;
    if ((p.vec_ok & 2) && nv >= 8) {
      float w2[4] = {v[4], v[5], v[6], v[7]};
      epilogue4(p, z, coff, m, n + 4, w2);
      return;
    }
  }
  for (int j = j0; j < 8; ++j)
    if (n + j < p.N) epilogue(p, z, coff, m, n + j, v[j]);
}


// ---- streamlined epilogue of a full-width tile whose f32 image sits in LDS ([rows][BN+4]).  One thread = 8 consecutive
// columns of ROWS_IT rows; the epilogue kind is a compile-time constant, bias is loaded once, all aux_in loads are issued
// before the first LDS read.  Same arithmetic as epilogue8() (bit-identical results).
template <int EPI, int ITERS, int ROW_STEP, int TW = BN, bool UNAL = false, bool PART = false>
__device__ __forceinline__ void fast_epilogue(const GemmP& p, const float* sC, int z, long long coff, int m_first, int n0,
                                              int row_l0) {
  constexpr int LDS_C = TW + 4;
  constexpr bool AUX_IN = EPI == EPI_RESID || EPI == EPI_DSWISH || EPI == EPI_MUL_POS;
  const int c8 = (threadIdx.x & (TW / 8 - 1)) * 8;
  const int n = n0 + c8;
  if (PART) {  // the matrix ends inside this tile (N % 4 == 0): a thread's chunk is whole, the last four columns, or outside
    if (n >= p.N) return;
    if (n + 8 > p.N) {
#pragma unroll 2
      for (int it = 0; it < ITERS; ++it) {
        const int m = m_first + it * ROW_STEP;
        if (m >= p.M) break;
        const float4 a = *reinterpret_cast<const float4*>(sC + (row_l0 + it * ROW_STEP) * LDS_C + c8);
        float w[4] = {a.x, a.y, a.z, a.w};
        epilogue4(p, z, coff, m, n, w);
      }
      return;
    }
  }
  float b8[8];
  if (p.bias) ld8x(p.bias, n, MI_DT_F32, b8);
  else {
#pragma unroll
    for (int j = 0; j < 8; ++j) b8[j] = 0.f;
  }
  const uint32_t dbase = (uint32_t)z * (uint32_t)(p.M * p.N) + (uint32_t)n;
  // Without a row map the rows of one thread are
  // ROW_STEP apart, so every address is (per-thread base, computed once) + it * (uniform stride, scalar unit) instead of a
  // 64-bit multiply-add per row and tensor (v_mad_u64_u32 + 2 x v_mul_lo_u32, quarter rate)
  // (the row map is only ever combined with EPI_MUL_POS -- the four conv2 input-gradient GEMMs; for every other epilogue
  // kind `lin` is a compile-time constant and the mapped path is not emitted at all).  Same-box A/B: -1 % on the FFN shapes.
  const bool lin = (EPI != EPI_MUL_POS) || !p.r_on;
  const long long ci_l = coff + (long long)m_first * p.ldc + n, ai_l = coff + (long long)m_first * p.ldaux + n;
  const long long ci_s = (long long)ROW_STEP * p.ldc, ai_s = (long long)ROW_STEP * p.ldaux;
  const uint32_t db_l = dbase + (uint32_t)m_first * (uint32_t)p.N, db_s = (uint32_t)ROW_STEP * (uint32_t)p.N;
  float aux[AUX_IN ? ITERS : 1][8];
  if (AUX_IN) {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int m = m_first + it * ROW_STEP;
      const long long ai0 = lin ? ai_l + it * ai_s : coff + crow(p, m) * p.ldaux + n;
      if (m < p.M) ld8x(p.aux_in, ai0, EPI == EPI_RESID ? MI_DT_F32 : p.auxin_dt, aux[it]);
    }
  }
  // every LDS read of the thread's rows goes out before the first use (rows past M are read and ignored; measured neutral to -0.1 ms per step,
  // profiles/r6_raw/epi_hoist_ab.txt)
  float4 ra[ITERS], rb[ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const float* src = sC + (row_l0 + it * ROW_STEP) * LDS_C + c8;
    if (TW == 256) {
      // 32 lanes per window row, 32 bytes apart: in the 16-lane service groups of ds_read_b128 ({0-3, 12-15, 20-27}, ...) both halves
      // of a lane pair would hit the same 16-byte slot twice (2-way) if every lane read its low half first; lanes 16-31 of each half
      // wave start with their high half instead -- every group then covers the 16 slots of the 256-byte bank row once
      const int hi = (threadIdx.x >> 4) & 1;
      const float4 x = *reinterpret_cast<const float4*>(src + 4 * hi);
      const float4 y = *reinterpret_cast<const float4*>(src + 4 * (1 - hi));
      ra[it] = hi ? y : x; rb[it] = hi ? x : y;
    } else {
      ra[it] = *reinterpret_cast<const float4*>(src);
      rb[it] = *reinterpret_cast<const float4*>(src + 4);
    }
  }
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int m = m_first + it * ROW_STEP;
    if (m >= p.M) continue;
    const float4 a = ra[it], b = rb[it];
    float v[8] = {a.x + b8[0], a.y + b8[1], a.z + b8[2], a.w + b8[3], b.x + b8[4], b.y + b8[5], b.z + b8[6], b.w + b8[7]};
    float dm[8];
    if (UNAL) drop_mask8u(p.drop, db_l + (uint32_t)it * db_s, dm);   // width 4 modulo 8: every other row starts inside a hash group
    else drop_mask8(p.drop, db_l + (uint32_t)it * db_s, dm);
    const long long mrow = lin ? 0 : crow(p, m);
    const long long ci = lin ? ci_l + it * ci_s : coff + mrow * p.ldc + n;
    const long long ai = lin ? ai_l + it * ai_s : coff + mrow * p.ldaux + n;
    if (EPI == EPI_STORE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= p.alpha * dm[j];
    } else if (EPI == EPI_SWISH_DROP) {
      if (p.swish_g) {
        float g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) swish_pair(v[j], dm[j], v[j], g[j]);
        st8x(p.aux_out, ai, p.auxout_dt, g);
      } else {
        st8x(p.aux_out, ai, p.auxout_dt, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = swishf_(v[j]) * dm[j];
      }
    } else if (EPI == EPI_RESID) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = aux[AUX_IN ? it : 0][j] + p.alpha * v[j] * dm[j];
    } else if (EPI == EPI_DSWISH) {
      if (p.swish_g) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= aux[AUX_IN ? it : 0][j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * dm[j] * swish_grad(aux[AUX_IN ? it : 0][j]);
      }
    } else if (EPI == EPI_RELU_MASK) {
      const int bb = m / p.rows_per_b;
      const int t = (m - bb * p.rows_per_b) / p.rows_inner;
      const bool ok = (long long)t < p.row_len[bb];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (ok && v[j] > 0.f) ? v[j] : 0.f;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = aux[AUX_IN ? it : 0][j] > 0.f ? v[j] : 0.f;
    }
#if defined(V8_EPI_ABL) && V8_EPI_ABL == 3
    asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(ci));
#else
    st8x(p.C, ci, p.c_dt, v);
#endif
  }
}
template <int ITERS, int ROW_STEP, int TW = BN, bool UNAL = false, bool PART = false>
__device__ __forceinline__ void fast_epilogue_any(const GemmP& p, const float* sC, int z, long long coff, int m_first, int n0,
                                                  int row_l0) {
  switch (p.epi) {
    case EPI_STORE: fast_epilogue<EPI_STORE, ITERS, ROW_STEP, TW, UNAL, PART>(p, sC, z, coff, m_first, n0, row_l0); break;
    case EPI_SWISH_DROP: fast_epilogue<EPI_SWISH_DROP, ITERS, ROW_STEP, TW, UNAL, PART>(p, sC, z, coff, m_first, n0, row_l0); break;
    case EPI_RESID: fast_epilogue<EPI_RESID, ITERS, ROW_STEP, TW, UNAL, PART>(p, sC, z, coff, m_first, n0, row_l0); break;
    case EPI_DSWISH: fast_epilogue<EPI_DSWISH, ITERS, ROW_STEP, TW, UNAL, PART>(p, sC, z, coff, m_first, n0, row_l0); break;
    case EPI_RELU_MASK: fast_epilogue<EPI_RELU_MASK, ITERS, ROW_STEP, TW, UNAL, PART>(p, sC, z, coff, m_first, n0, row_l0); break;
    default: fast_epilogue<EPI_MUL_POS, ITERS, ROW_STEP, TW, UNAL, PART>(p, sC, z, coff, m_first, n0, row_l0); break;
  }
}

// =================================================================================================
// bf16 MFMA kernel
// =================================================================================================

__device__ __forceinline__ int lds_off(int r, int chunk) {  // element offset of 8-element chunk `chunk` of row r
  return r * BK + ((chunk ^ ((r >> 1) & 7)) << 3);
}

// ---- staging of a [rows][K] (K-contiguous) operand tile: 128 rows x 64 k = 1024 16-B chunks, 4 per thread
struct StageN { u32x4 v[4]; };
__device__ __forceinline__ void load_n(StageN& s, const bf16_t* base, long long ld, int row0, int rows, int k0, int K) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = threadIdx.x + i * 256;
    const int r = q >> 3, ck = q & 7;
    const int gr = row0 + r, gk = k0 + ck * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (gr < rows && gk < K) v = *reinterpret_cast<const u32x4*>(base + (long long)gr * ld + gk);
    s.v[i] = v;
  }
}
__device__ __forceinline__ void store_n(const StageN& s, bf16_t* lds) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = threadIdx.x + i * 256;
    const int r = q >> 3, ck = q & 7;
    *reinterpret_cast<u32x4*>(lds + lds_off(r, ck)) = s.v[i];
  }
}
// ---- LDS-DMA staging of a [rows][K] operand tile (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR / ds_write
// round trip).  The LDS image is lane-linear, so the bank swizzle is applied to the SOURCE chunk: LDS[r][c] <- global
// chunk c ^ ((r>>1)&7); the fragment reads use the same involution (lds_off).  Rows past the matrix are clamped (their
// results are never stored); k-chunks past K read a zero page.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
__device__ __attribute__((aligned(16))) uint32_t g_zero16[4] = {0u, 0u, 0u, 0u};
__device__ __forceinline__ void glds_n(const bf16_t* base, long long ld, int row0, int rows, int k0, int K, bf16_t* lds_tile) {
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = threadIdx.x + i * 256;
    const int r = q >> 3, ck = q & 7;
    const int gck = ck ^ ((r >> 1) & 7);
    int gr = row0 + r;
    gr = gr < rows ? gr : rows - 1;
    const int gk = k0 + gck * 8;
    const bf16_t* src = (gk < K) ? base + (long long)gr * ld + gk : reinterpret_cast<const bf16_t*>(g_zero16);
    bf16_t* dst = lds_tile + (wave * 64 + i * 256) * 8;  // wave-uniform; lane l lands at dst + 16*l bytes
    __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)dst, 16, 0, 0);
  }
}
// ---- staging of a [K][rows] (reduction-major) operand tile: 64 k-rows x 128 cols; thread = (k-group of 4, 8-col chunk)
struct StageT { u32x4 v[4]; };
__device__ __forceinline__ void load_t(StageT& s, const bf16_t* base, long long ld, int col0, int cols, int k0, int K) {
  const int g = threadIdx.x & 15, c = threadIdx.x >> 4;
  const int gc = col0 + c * 8;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int gk = k0 + g * 4 + j;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (gk < K && gc < cols) v = *reinterpret_cast<const u32x4*>(base + (long long)gk * ld + gc);
    s.v[j] = v;
  }
}
__device__ __forceinline__ void store_t(const StageT& s, bf16_t* lds) {
  const int g = threadIdx.x & 15, c = threadIdx.x >> 4;
#pragma unroll
  for (int e = 0; e < 8; ++e) {  // column 8c+e holds k = 4g..4g+3
    const int w = e >> 1, hi = e & 1;
    uint32_t x0 = s.v[0][w], x1 = s.v[1][w], x2 = s.v[2][w], x3 = s.v[3][w];
    uint32_t lo01, lo23;
    if (hi) { lo01 = (x0 >> 16) | (x1 & 0xffff0000u); lo23 = (x2 >> 16) | (x3 & 0xffff0000u); }
    else    { lo01 = (x0 & 0xffffu) | (x1 << 16);     lo23 = (x2 & 0xffffu) | (x3 << 16); }
    const int r = c * 8 + e;
    const int k = g * 4;
    u32x2 o = {lo01, lo23};
    *reinterpret_cast<u32x2*>(lds + lds_off(r, k >> 3) + (k & 7)) = o;
  }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmP p) {
  drop_resolve(p.drop);
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * (BM + BN) * BK];  // 64 KiB
  // layout: A buf0 | A buf1 | B buf0 | B buf1
#define SA(buf) (smem + (buf) * (BM * BK))
#define SB(buf) (smem + 2 * BM * BK + (buf) * (BN * BK))

  // ---- XCD-aware tile mapping (bijective)
  const int tn = (p.N + BN - 1) / BN, tm = (p.M + BM - 1) / BM;
  const int ntiles = tm * tn;
  const int bid = blockIdx.x;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int tile_m = logical / tn, tile_n = logical - tile_m * tn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int z = blockIdx.z;
  const int z0 = z % p.nb0, z1 = z / p.nb0;
  const bf16_t* A = (const bf16_t*)p.A + z0 * p.sA0 + z1 * p.sA1;
  const bf16_t* B = (const bf16_t*)p.B + z0 * p.sB0 + z1 * p.sB1;
  const long long coff = z0 * p.sC0 + z1 * p.sC1;

  const int nk_total = (p.K + BK - 1) / BK;
  int kt_begin = 0, kt_end = nk_total;
  if (p.splitk > 1) {
    kt_begin = blockIdx.y * p.ktiles_per_split;
    kt_end = min(nk_total, kt_begin + p.ktiles_per_split);
    if (kt_begin >= kt_end) return;
  }

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // K-contiguous operands go global -> LDS by DMA (glds_n); reduction-major operands are transposed through registers.
  StageT at, bt;
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool do_colsum = TA && p.colsum_out != nullptr && tile_n == 0;
  auto issue_loads = [&](int kt, int buf) {
    const int k0 = kt * BK;
    if (TA) load_t(at, A, p.lda, m0, p.M, k0, p.K); else glds_n(A, p.lda, m0, p.M, k0, p.K, SA(buf));
    if (TB) load_t(bt, B, p.ldb, n0, p.N, k0, p.K); else glds_n(B, p.ldb, n0, p.N, k0, p.K, SB(buf));
  };
  auto finish_loads = [&](int buf) {
    if (TA && do_colsum) {  // this thread holds A rows k = 4g..4g+3, columns 8c..8c+7 of the tile (zero-filled outside)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          csum[2 * w] += __uint_as_float(at.v[j][w] << 16);
          csum[2 * w + 1] += __uint_as_float(at.v[j][w] & 0xffff0000u);
        }
    }
    if (TA) store_t(at, SA(buf));
    if (TB) store_t(bt, SB(buf));
    if (!TA || !TB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  issue_loads(kt_begin, 0);
  finish_loads(0);
  __syncthreads();

  const int lr = lane & 31, lh = lane >> 5;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int cur = (kt - kt_begin) & 1;
    const bool more = (kt + 1 < kt_end);
    if (more) issue_loads(kt + 1, cur ^ 1);
    const bf16_t* a_s = SA(cur);
    const bf16_t* b_s = SB(cur);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + lr;
        af[i] = *reinterpret_cast<const bf16x8*>(a_s + lds_off(ra, kk * 2 + lh));
        const int rb = wn * 64 + i * 32 + lr;
        bfr[i] = *reinterpret_cast<const bf16x8*>(b_s + lds_off(rb, kk * 2 + lh));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (more) finish_loads(cur ^ 1);
    __syncthreads();
  }

  if (TA && do_colsum) {  // reduce over the 16 k-groups (16 consecutive lanes share the column chunk), 1 atomic / column
    const int c = threadIdx.x >> 4;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = csum[e];
      v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
      const int m = m0 + c * 8 + e;
      if ((threadIdx.x & 15) == 0 && m < p.M) atomicAdd(p.colsum_out + z0 * p.colsum_stride + m, v);
    }
  }

  // ---- epilogue through LDS: accumulators (C/D layout of the 32x32 MFMA: col = lane&31,
  // row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) are transposed into row-major f32 tiles of 64 x 128 so that every
  // thread owns 8 consecutive columns of a row: bias / aux loads and the C store are 16-32 B per lane, 256-512 B per row.
  constexpr int LDS_C = BN + 4;
  float* sC = reinterpret_cast<float*>(smem);  // 64 x 132 floats = 33 KiB (the staging buffers are dead now)
  for (int half = 0; half < 2; ++half) {
    if (half) __syncthreads();
    if (wm == half) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row_l = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            sC[row_l * LDS_C + wn * 64 + j * 32 + lr] = acc[i][j][r];
          }
    }
    __syncthreads();
    if (p.atomic) {
      // split-K accumulation: lane-contiguous columns so that each atomic instruction covers whole 256-B row segments
      for (int e = threadIdx.x; e < 64 * BN; e += 256) {
        const int row_l = e >> 7, col = e & (BN - 1);
        const int m = m0 + half * 64 + row_l, n = n0 + col;
        if (m < p.M && n < p.N) epilogue(p, z, coff, m, n, sC[row_l * LDS_C + col]);
      }
      continue;
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row_l = (threadIdx.x >> 4) + 16 * it;
      const int c8 = (threadIdx.x & 15) * 8;
      const int m = m0 + half * 64 + row_l, n = n0 + c8;
      if (m < p.M && n < p.N) {
        float v[8];
        const float4 a = *reinterpret_cast<const float4*>(sC + row_l * LDS_C + c8);
        const float4 b = *reinterpret_cast<const float4*>(sC + row_l * LDS_C + c8 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        epilogue_chunk(p, z, coff, m, n, v);
      }
    }
  }
}

// =================================================================================================
// Second structure (large GEMMs, all layouts): 256x128x64 block tile, 8 waves (4x2, each 64x64), THREE LDS stages filled
// by LDS-DMA two K-tiles ahead, one raw s_barrier per K-tile with a COUNTED vmcnt (the newest tile's DMA stays in flight
// across the barrier; __syncthreads() would drain it).  144 KiB LDS -> one workgroup per CU, two waves per SIMD.
//   * K-contiguous operand  ([rows][K]):  LDS image [rows][64] (128-B rows), chunk swizzle c ^= (row>>1)&7, ds_read_b128.
//   * reduction-major operand ([K][rows], wgrad / P.V): LDS image [64][rows] exactly as in memory (so it can be DMA'd),
//     chunk swizzle c ^= (k&3)<<2, fragments fetched with ds_read_b64_tr_b16 -- the hardware transpose read: lane t of
//     a 16-lane group supplies &img[k0 + t/4][col0 + (t%4)*4] and receives img[k0..k0+3][col0 + t] (probed on gfx950,
//     tools/probe/tr_probe.hip).  No register transpose, no ds_write.
// =================================================================================================
#define BM2 256
#define NT2_STAGE ((BM2 + BN) * BK)  // elements per stage (A 256x64 | B 128x64) = 48 KiB
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

// Per-thread LDS-DMA sources of one operand, computed once per tile: the K loop only adds the k-tile step.
template <int R> struct DmaSrc {
  static constexpr int PER = R * 8 / 512;
  const bf16_t* ptr[PER];  // source of chunk i at k-tile 0; a chunk outside the matrix points at the zero page ...
  int kin[PER];            // k offset of the chunk inside a k-tile (K-tail test, last k-tile only)
  int step[PER];           // ... and advances by 0 elements per k-tile instead of BK (* ld): the issue has no select
};
template <bool T, int R>
__device__ __forceinline__ void dma_setup(DmaSrc<R>& d, const bf16_t* base, long long ld, int row0, int rows, int kbase) {
#pragma unroll
  for (int i = 0; i < DmaSrc<R>::PER; ++i) {
    const int q = threadIdx.x + i * 512;
    if (!T) {
      const int r = q >> 3, ck = q & 7;
      const int gck = ck ^ ((r >> 1) & 7);
      int gr = row0 + r;
      gr = gr < rows ? gr : rows - 1;
      d.ptr[i] = base + (long long)gr * ld + kbase + gck * 8;
      d.kin[i] = gck * 8;
      d.step[i] = BK;
    } else {
      constexpr int CPR = R / 8;
      const int k = q / CPR, cp = q % CPR;
      const int c = cp ^ ((k & 3) << 2);
      const int gc = row0 + c * 8;
      const bool ok = gc < rows;
      d.ptr[i] = ok ? base + (long long)(kbase + k) * ld + gc : reinterpret_cast<const bf16_t*>(g_zero16);
      d.kin[i] = k;
      d.step[i] = ok ? (int)(BK * ld) : 0;  // BK * ld < 2^31 is checked by the host entry
    }
  }
}
// implicit-GEMM gather of a K-contiguous operand: the chunk's row is a conv output position, its pointer the channel
// vector at the position's origin; per K-tile only the (uniform) tap offset and the bounds test change
template <int R> struct DmaGather {
  static constexpr int PER = R * 8 / 512;
  const bf16_t* org[PER];  // &src[b][i*si][j*sj][chunk*8]
  int gi[PER], gj[PER];    // i*si, j*sj
};
template <int R>
__device__ __forceinline__ void gather_setup(DmaGather<R>& g, const GemmP& p, const bf16_t* base, int row0) {
#pragma unroll
  for (int i = 0; i < DmaGather<R>::PER; ++i) {
    const int q = threadIdx.x + i * 512;
    const int r = q >> 3, ck = q & 7;
    const int gck = ck ^ ((r >> 1) & 7);
    int m = row0 + r;
    m = m < p.M ? m : p.M - 1;
    const int per_b = p.g_nI * p.g_nJ;
    const int b = m / per_b, rr = m - b * per_b;
    const int oi = rr / p.g_nJ, oj = rr - oi * p.g_nJ;
    g.gi[i] = oi * p.g_si; g.gj[i] = oj * p.g_sj;
    g.org[i] = base + (((long long)b * p.g_SI + g.gi[i]) * p.g_SJ + g.gj[i]) * p.g_C + gck * 8;
  }
}
template <int R>
__device__ __forceinline__ void gather_issue(const DmaGather<R>& g, const GemmP& p, int kt, bf16_t* lds_tile) {
  const int wave = threadIdx.x >> 6;
  const int k0 = kt * BK;
  const int tap = k0 / p.g_C, c0 = k0 - tap * p.g_C;  // uniform; C % 64 == 0 keeps a K-tile inside one tap
  const int di = tap_delta(p.g_dip, tap), dj = tap_delta(p.g_djp, tap);
  const long long toff = ((long long)di * p.g_SJ + dj) * p.g_C + c0;
#pragma unroll
  for (int i = 0; i < DmaGather<R>::PER; ++i) {
    const bool ok = (unsigned)(g.gi[i] + di) < (unsigned)p.g_SI && (unsigned)(g.gj[i] + dj) < (unsigned)p.g_SJ;
    const bf16_t* src = ok ? g.org[i] + toff : reinterpret_cast<const bf16_t*>(g_zero16);
    bf16_t* dst = lds_tile + (wave * 64 + i * 512) * 8;
    __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)dst, 16, 0, 0);
  }
}

// implicit-GEMM gather of a REDUCTION-MAJOR operand (the conv weight gradient: K runs over the output positions, the
// operand's columns are the input channels of one tap).  A chunk is (k-row of the K-tile, 8 channels); its position
// (b, i, j) advances by 64 rows per K-tile and is updated incrementally.
template <int R> struct DmaGatherT {
  static constexpr int PER = R * 8 / 512;
  int i[PER], j[PER];          // output position (within its utterance) of the chunk's row at the current K-tile
  int off[PER];                // element offset of the position's origin src[b][i*si][j*sj][col] (< 2^31 by contract)
  int row[PER];                // the row index m itself (K-tail test)
  bool colok[PER];
};
template <int R>
__device__ __forceinline__ void gatherT_setup(DmaGatherT<R>& g, const GemmP& p, int col0, int ncols, int kbase) {
  constexpr int CPR = R / 8;
#pragma unroll
  for (int c = 0; c < DmaGatherT<R>::PER; ++c) {
    const int q = threadIdx.x + c * 512;
    const int k = q / CPR, cp = q % CPR;
    const int ch = cp ^ ((k & 3) << 2);
    const int gc = col0 + ch * 8;
    g.colok[c] = gc < ncols;
    const int m = kbase + k;
    const int per_b = p.g_nI * p.g_nJ;
    const int b = m / per_b;
    const int rr = m - b * per_b;
    g.i[c] = rr / p.g_nJ;
    g.j[c] = rr - g.i[c] * p.g_nJ;
    g.row[c] = m;
    g.off[c] = ((b * p.g_SI + g.i[c] * p.g_si) * p.g_SJ + g.j[c] * p.g_sj) * p.g_C + (g.colok[c] ? gc : 0);
  }
}
template <int R>
__device__ __forceinline__ void gatherT_issue(DmaGatherT<R>& g, const GemmP& p, const bf16_t* base, int tap,
                                              bf16_t* lds_tile, const bool load = true /* uniform; false: only advance the positions */) {
  const int wave = threadIdx.x >> 6;
  const int di = tap_delta(p.g_dip, tap), dj = tap_delta(p.g_djp, tap);
  const int toff = (di * p.g_SJ + dj) * p.g_C;                                   // uniform
  const int step_j = BK * p.g_sj * p.g_C;                                        // one K-tile = 64 positions further
  const int wrap_j = p.g_si * p.g_SJ * p.g_C - p.g_nJ * p.g_sj * p.g_C;          // j -= nJ, i += 1
  const int wrap_i = p.g_SI * p.g_SJ * p.g_C - p.g_nI * p.g_si * p.g_SJ * p.g_C; // i -= nI, b += 1
#pragma unroll
  for (int c = 0; c < DmaGatherT<R>::PER; ++c) {
    const int si = g.i[c] * p.g_si + di, sj = g.j[c] * p.g_sj + dj;
    const bool ok = g.colok[c] && g.row[c] < p.K && (unsigned)si < (unsigned)p.g_SI && (unsigned)sj < (unsigned)p.g_SJ;
    const bf16_t* src = ok ? base + (g.off[c] + toff) : reinterpret_cast<const bf16_t*>(g_zero16);
    bf16_t* dst = lds_tile + (wave * 64 + c * 512) * 8;
    if (load) __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)dst, 16, 0, 0);
    // advance by one K-tile
    g.row[c] += BK; g.j[c] += BK; g.off[c] += step_j;
    while (g.j[c] >= p.g_nJ) { g.j[c] -= p.g_nJ; ++g.i[c]; g.off[c] += wrap_j; }
    while (g.i[c] >= p.g_nI) { g.i[c] -= p.g_nI; g.off[c] += wrap_i; }
  }
}

template <int R>
__device__ __forceinline__ void dma_issue(const DmaSrc<R>& d, int it, int k0, int K, bool ktail, bf16_t* lds_tile) {
  const int wave = threadIdx.x >> 6;
  const bool tail = ktail && (k0 + BK > K);  // uniform: only the last k-tile of a ragged K tests its chunks
  const unsigned long long zero = (unsigned long long)(const void*)g_zero16;
#pragma unroll
  for (int i = 0; i < DmaSrc<R>::PER; ++i) {
    unsigned long long src = (unsigned long long)(const void*)(d.ptr[i] + (long long)it * d.step[i]);
    if (tail) src = (k0 + d.kin[i] < K) ? src : zero;  // integer select (v_cndmask), not a divergent branch
    bf16_t* dst = lds_tile + (wave * 64 + i * 512) * 8;
    __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)dst, 16, 0, 0);
  }
}

template <bool T, int R>
__device__ __forceinline__ bf16x8 frag_v2(const bf16_t* tile, int row_tile, int kk, int lane) {
  if (!T) {
    return *reinterpret_cast<const bf16x8*>(tile + lds_off(row_tile + (lane & 31), kk * 2 + (lane >> 5)));
  } else {
    const int t = lane & 15, g4 = (lane >> 4) & 1, lh = lane >> 5;
    const int col = row_tile + g4 * 16 + (t & 3) * 4;
    union { bf16x8 v; s16x4 h[2]; } u;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int krow = kk * 16 + lh * 8 + r * 4 + (t >> 2);
      const int off = krow * R + ((((col >> 3) ^ ((t >> 2) << 2))) << 3) + (col & 7);
      u.h[r] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(tile + off));
    }
    return u.v;
  }
}

// ---- transpose reads by inline asm, for the loops whose operands are BOTH reduction-major (weight gradients).
// hipcc puts `s_waitcnt vmcnt(0)` in front of a ds_read_b64_tr_b16 it emits itself whenever an LDS-DMA is in flight (it
// cannot tell the stages apart), i.e. right after the prefetch of the next K-tiles has been issued: the counted vmcnt of
// these loops never got to do its job and every K-tile paid a full memory latency.  Issued by hand the reads carry no such
// wait; the price is that the compiler does not know when their destinations land -- every use sits behind an inline
// `s_waitcnt lgkmcnt` that names the registers ("+v"), and tools/check_asm_loads.py lints the ISA for stray copies.
typedef __attribute__((address_space(3))) const char lds_cchar_t;
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(lds_cchar_t*)p; }
union FragU { bf16x8 v; s16x4 h[2]; };
template <int OFF>
__device__ __forceinline__ void tr_rd(s16x4& dst, uint32_t addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF) : "memory");
}
// LDS byte address of the (kk = 0, r = 0) read of frag_v2<true, R>(tile, row_tile, ...): the swizzle term depends on
// krow & 3 = (lane & 15) >> 2 only, so k-step kk / half r are the compile-time offset (kk * 16 + r * 4) * R * 2
template <int R>
__device__ __forceinline__ uint32_t tr_base(const bf16_t* tile, int row_tile, int lane) {
  const int t = lane & 15, g4 = (lane >> 4) & 1, lh = lane >> 5;
  const int col = row_tile + g4 * 16 + (t & 3) * 4;
  const int krow = lh * 8 + (t >> 2);
  return lds_addr(tile) + (uint32_t)((krow * R + ((((col >> 3) ^ ((t >> 2) << 2))) << 3) + (col & 7)) * 2);
}
template <int R, int KK>
__device__ __forceinline__ void tr_frag_rd(FragU& f, uint32_t base) {
  tr_rd<(KK * 16) * R * 2>(f.h[0], base);
  tr_rd<(KK * 16 + 4) * R * 2>(f.h[1], base);
}
// bias-gradient read of the staged A tile (8 bytes), self-contained (issue + wait)
__device__ __forceinline__ u32x2 lds_rd64_sync(uint32_t addr) {
  u32x2 v;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

// body of the 256x128 structure: workgroup `bid` of the problem's tile grid, K slice `kslice`, batch index `z`
template <bool TA, bool TB, bool REG = false>
__device__ __forceinline__ void gemm_v2_body(const GemmP& p, bf16_t* smem2, const int bid, const int kslice, const int z
                                             , const bool direct = false
) {
  const int tn = (p.N + BN - 1) / BN, tm = (p.M + BM2 - 1) / BM2;
  const int ntiles = tm * tn;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int logical = direct ? bid : (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int tile_m = logical / tn, tile_n = logical - tile_m * tn;
  const int m0 = tile_m * BM2, n0 = tile_n * BN;
  const int z0 = z % p.nb0, z1 = z / p.nb0;
  const bf16_t* A = (const bf16_t*)p.A + z0 * p.sA0 + z1 * p.sA1;
  const bf16_t* B = (const bf16_t*)p.B + z0 * p.sB0 + z1 * p.sB1;
  const long long coff = z0 * p.sC0 + z1 * p.sC1;
  const int nk_total = (p.K + BK - 1) / BK;
  int kt0 = 0, kt1 = nk_total;
  if (p.splitk > 1) {
    kt0 = kslice * p.ktiles_per_split;
    kt1 = min(nk_total, kt0 + p.ktiles_per_split);
    if (kt0 >= kt1) return;
  }
#ifdef GEMM_ABLATE
  const int dbg = p.vec_ok >> 8;
  const int nk = (dbg & 2) ? 0 : kt1 - kt0;
#else
  const int nk = kt1 - kt0;
#endif
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  DmaSrc<BM2> dA;
  DmaSrc<BN> dB;
  DmaGather<BM2> gA;
  const bool gatherA = !TA && p.g_on == 1;
  if (gatherA) gather_setup<BM2>(gA, p, A, m0);
  else dma_setup<TA, BM2>(dA, A, p.lda, m0, p.M, kt0 * BK);
  DmaGatherT<BN> gB;
  const bool gatherB = TB && p.g_on == 2;  // tap = batch index z0; p.B is the un-batched source grid
  if (gatherB) gatherT_setup<BN>(gB, p, n0, p.N, kt0 * BK);
  else dma_setup<TB, BN>(dB, B, p.ldb, n0, p.N, kt0 * BK);
  const bool ktail = (p.K & (BK - 1)) != 0;
  // it = local tile index (issued in increasing order: the gathers advance incrementally).  (Measured: issuing the two
  // operands separately BETWEEN the MFMA groups of the tile being multiplied, instead of as one block behind the K-tile
  // barrier where all eight waves are in phase, is 2-6 % slower on every Conformer shape -- profiles/r2_gemm_structures.md 6.)
  auto issue_a = [&](int it) {
    bf16_t* st = smem2 + (it % 3) * NT2_STAGE;
    if (gatherA) gather_issue<BM2>(gA, p, kt0 + it, st);
    else dma_issue<BM2>(dA, it, (kt0 + it) * BK, p.K, ktail, st);
  };
  auto issue_b = [&](int it) {
    bf16_t* st = smem2 + (it % 3) * NT2_STAGE;
    if (gatherB) gatherT_issue<BN>(gB, p, (const bf16_t*)p.B, z0, st + BM2 * BK);
    else dma_issue<BN>(dB, it, (kt0 + it) * BK, p.K, ktail, st + BM2 * BK);
  };
  auto issue = [&](int it) { issue_a(it); issue_b(it); };
  // bias gradient riding along with wgrad: column sums of the A tile, read back from LDS (8 B per lane per k-row group)
  // The tn workgroups of one (tile_m, K slice) stage the same A tile: its 64 k-rows are dealt round-robin to (up to 8 of)
  // them, so the column-sum work is spread evenly instead of making the tile_n == 0 workgroups the stragglers.
  const int cs_step = tn < 8 ? tn : 8;
  const bool do_colsum = TA && p.colsum_out != nullptr && tile_n < cs_step;
  float csum[4] = {0.f, 0.f, 0.f, 0.f};

  // ---- K loop.  Two schedules, chosen per operand layout by measurement (same box, old | new, us):
  //   * both operands reduction-major (weight gradients; 32 ds_read_b64_tr_b16 per K-tile): software-pipelined at half-K-tile
  //     granularity -- the reads are two bursts of 8 fragments (k-steps 0-1 | 2-3), each issued right before the 8 MFMAs of
  //     the OTHER half, so they run under matrix work inside every wave: FFN wgrad 77.2 -> 70.4, conv2-like wgrad 869 -> 805;
  //   * otherwise (16 ds_read_b128 per K-tile): all fragment reads of the K-tile in ONE burst, then the 16 MFMAs -- the
  //     LDS, not the MFMA pipe, bounds this loop and only reaches its rate on long bursts (pipelined: FFN2 forward 53.8 ->
  //     56.9, 4096^3 NN 140.7 -> 147.6).
  constexpr bool PIPELINED = TA && TB;
  if constexpr (REG) {
  // Register-prefetch schedule (see the sixth structure below for the why): K-contiguous operands, nk even and >= 4, dense
  // operands.  Two LDS stages of this structure's three; every thread carries its 4 + 2 chunks of two future K-tiles, copies one
  // tile per step into the idle stage (ds_write_b128, spread over the step's k-steps) and reloads the registers for the tile
  // three steps ahead.
  static_assert(!REG || (!TA && !TB), "register prefetch: K-contiguous operands");
  const char* Ab = (const char*)A + (long long)kt0 * (BK * 2);
  const char* Bb = (const char*)B + (long long)kt0 * (BK * 2);
  uint32_t oa[4], ob[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = threadIdx.x + i * 512;
    const int r = q >> 3, ck = q & 7;
    const int gck = ck ^ ((r >> 1) & 7);
    int ga = m0 + r; ga = ga < p.M ? ga : p.M - 1;
    oa[i] = (uint32_t)((long long)ga * p.lda + gck * 8) * 2u;
    if (i < 2) {
      int gb = n0 + r; gb = gb < p.N ? gb : p.N - 1;
      ob[i] = (uint32_t)((long long)gb * p.ldb + gck * 8) * 2u;
    }
  }
  struct RT { u32x4 a[4]; u32x4 b[2]; };
  auto ld_tile = [&](RT& t, int it) {
    const bool in = it < nk;
    const uint32_t msk = in ? 0xffffffffu : 0u;
    const char* a = in ? Ab + (long long)it * (BK * 2) : reinterpret_cast<const char*>(g_zero16);
    const char* b = in ? Bb + (long long)it * (BK * 2) : reinterpret_cast<const char*>(g_zero16);
#pragma unroll
    for (int i = 0; i < 4; ++i) t.a[i] = *reinterpret_cast<const u32x4*>(a + (oa[i] & msk));
#pragma unroll
    for (int i = 0; i < 2; ++i) t.b[i] = *reinterpret_cast<const u32x4*>(b + (ob[i] & msk));
  };
  auto st_tile = [&](const RT& t, int stage) {
    bf16_t* st = smem2 + stage * NT2_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(st + (threadIdx.x + i * 512) * 8) = t.a[i];
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(st + BM2 * BK + (threadIdx.x + i * 512) * 8) = t.b[i];
  };
  auto step = [&](RT& t, int wr, int rd, int next_it) {
    const bf16_t* a_s = smem2 + rd * NT2_STAGE;
    const bf16_t* b_s = a_s + BM2 * BK;
    bf16_t* st = smem2 + wr * NT2_STAGE;
    const bool in = next_it < nk;
    const uint32_t msk = in ? 0xffffffffu : 0u;
    const char* ga = in ? Ab + (long long)next_it * (BK * 2) : reinterpret_cast<const char*>(g_zero16);
    const char* gb = in ? Bb + (long long)next_it * (BK * 2) : reinterpret_cast<const char*>(g_zero16);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = frag_v2<false, BM2>(a_s, wm * 64 + i * 32, kk, lane);
        bfr[i] = frag_v2<false, BN>(b_s, wn * 64 + i * 32, kk, lane);
      }
      if (kk < 2) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int i = kk * 2 + c;
          *reinterpret_cast<u32x4*>(st + (threadIdx.x + i * 512) * 8) = t.a[i];
          t.a[i] = *reinterpret_cast<const u32x4*>(ga + (oa[i] & msk));
        }
      } else if (kk == 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          *reinterpret_cast<u32x4*>(st + BM2 * BK + (threadIdx.x + i * 512) * 8) = t.b[i];
          t.b[i] = *reinterpret_cast<const u32x4*>(gb + (ob[i] & msk));
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  RT r0, r1;
  ld_tile(r0, 0);
  ld_tile(r1, 1);
  st_tile(r0, 0);
  ld_tile(r0, 2);
  __syncthreads();
  for (int it = 0; it < nk; it += 2) {
    step(r1, 1, 0, it + 3);
    __syncthreads();
    step(r0, 0, 1, it + 4);
    __syncthreads();
  }
  } else
  if constexpr (PIPELINED) {
  // Pipelined schedule:
  //   on entry to iteration it:  F[0..1] = first half of tile it (read during iteration it-1, after its barrier)
  //   A  issue the DMA of tile it+2 (stage of tile it-1: its last reads completed before the barrier of iteration it-1)
  //   B  read the second half of tile it -> F[2..3]          C  MFMAs of F[0..1]
  //   D  my share of tile it+1 has landed (counted vmcnt), my reads are done (lgkmcnt 0), barrier -> tile it+1 visible
  //   E  read the first half of tile it+1 -> F[0..1]         F  MFMAs of F[2..3]
  FragU af[4][2], bfr[4][2];
  uint32_t abase[2], bbase[2];  // per-lane read addresses in stage 0 (inline-asm reads: see tr_rd)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    abase[i] = tr_base<BM2>(smem2, wm * 64 + i * 32, lane);
    bbase[i] = tr_base<BN>(smem2 + BM2 * BK, wn * 64 + i * 32, lane);
  }
#define V2_RD_HALF(SB, K0, K1)                                                                     \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                  \
    tr_frag_rd<BM2, K0>(af[K0][i], abase[i] + (SB)); tr_frag_rd<BN, K0>(bfr[K0][i], bbase[i] + (SB)); \
    tr_frag_rd<BM2, K1>(af[K1][i], abase[i] + (SB)); tr_frag_rd<BN, K1>(bfr[K1][i], bbase[i] + (SB)); \
  }
#define V2_WAIT_HALF(K0, K1)                                                                                   \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                          \
               : "+v"(af[K0][0].v), "+v"(af[K0][1].v), "+v"(bfr[K0][0].v), "+v"(bfr[K0][1].v), "+v"(af[K1][0].v), \
                 "+v"(af[K1][1].v), "+v"(bfr[K1][0].v), "+v"(bfr[K1][1].v)                                      \
               :                                                                                               \
               : "memory")
  auto mfma_half = [&](const int h) {
#pragma unroll
    for (int kk = 2 * h; kk < 2 * h + 2; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk][i].v, bfr[kk][j].v, acc[i][j], 0, 0, 0);
  };
  if (nk > 0) {
    issue(0);
    if (nk > 1) issue(1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    V2_RD_HALF(0u, 0, 1);
    V2_WAIT_HALF(0, 1);
  }
  for (int it = 0; it < nk; ++it) {
    if (it + 2 < nk) issue(it + 2);
    const uint32_t sb = (uint32_t)((it % 3) * (NT2_STAGE * 2));
    __builtin_amdgcn_sched_barrier(0);
    V2_RD_HALF(sb, 2, 3);
    __builtin_amdgcn_sched_barrier(0);  // keep the burst ahead of the MFMAs: the scheduler would sink the reads to their uses
    mfma_half(0);
    if (TA && do_colsum) {  // thread -> (k-group of 8 rows = wave, 4 consecutive columns = lane): 8 x ds_read_b64
      const int col = lane * 4;
      const uint32_t a_addr = lds_addr(smem2) + sb;
      for (int kr = tile_n; kr < 8; kr += cs_step) {
        const int krow = wave * 8 + kr;
        const int off = krow * BM2 + (((col >> 3) ^ ((krow & 3) << 2)) << 3) + (col & 7);
        const u32x2 v = lds_rd64_sync(a_addr + (uint32_t)(off * 2));
        csum[0] += __uint_as_float(v[0] << 16); csum[1] += __uint_as_float(v[0] & 0xffff0000u);
        csum[2] += __uint_as_float(v[1] << 16); csum[3] += __uint_as_float(v[1] & 0xffff0000u);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // tile it+1 must have landed (this wave's share) before the barrier publishes it; tile it+2 may stay in flight
    if (it + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    V2_WAIT_HALF(2, 3);
    __builtin_amdgcn_s_barrier();
    // first half of tile it+1 (after the last tile: a harmless read of an idle stage -- an unconditional issue site keeps the
    // compiler from merging the destinations through copies)
    V2_RD_HALF((uint32_t)(((it + 1) % 3) * (NT2_STAGE * 2)), 0, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_half(1);
    __builtin_amdgcn_sched_barrier(0);  // (the wait is no scheduling barrier for the MFMAs: it would be hoisted above them)
    V2_WAIT_HALF(0, 1);  // landed under the 8 MFMAs above; nothing asm-loaded is in flight across the back edge
  }
#undef V2_RD_HALF
#undef V2_WAIT_HALF

  } else {
#ifdef GEMM_ABLATE
  if (nk > 0)
#endif
  issue(0);
  if (nk > 1) issue(1);
  if (nk > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (int it = 0; it < nk; ++it) {
#ifdef GEMM_ABLATE
    if (!(dbg & 8))
#endif
    if (it + 2 < nk) issue(it + 2);  // overwrites the stage of tile it-1: every wave passed the barrier after reading it
    const bf16_t* a_s = smem2 + (it % 3) * NT2_STAGE;
    const bf16_t* b_s = a_s + BM2 * BK;
    // all 16 fragment reads of the K-tile go out back to back (the LDS only reaches its rate on long bursts); the
    // MFMAs of step kk then wait for exactly their operands (in-order returns -> counted lgkmcnt)
    bf16x8 af[4][2], bfr[4][2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[kk][i] = frag_v2<TA, BM2>(a_s, wm * 64 + i * 32, kk, lane);
        bfr[kk][i] = frag_v2<TB, BN>(b_s, wn * 64 + i * 32, kk, lane);
      }
    __builtin_amdgcn_sched_barrier(0);  // keep the burst: the scheduler would sink the reads back next to their MFMAs
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#ifdef GEMM_ABLATE
      if (dbg & 4) {
#pragma unroll
        for (int i = 0; i < 2; ++i) { asm volatile("" :: "v"(af[kk][i])); asm volatile("" :: "v"(bfr[kk][i])); }
      } else
#endif
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk][i], bfr[kk][j], acc[i][j], 0, 0, 0);
    }
    if (TA && do_colsum) {  // thread -> (k-group of 8 rows = wave, 4 consecutive columns = lane): 8 x ds_read_b64
      const int col = lane * 4;
      for (int kr = tile_n; kr < 8; kr += cs_step) {
        const int krow = wave * 8 + kr;
        const int off = krow * BM2 + (((col >> 3) ^ ((krow & 3) << 2)) << 3) + (col & 7);
        const u32x2 v = *reinterpret_cast<const u32x2*>(a_s + off);
        csum[0] += __uint_as_float(v[0] << 16); csum[1] += __uint_as_float(v[0] & 0xffff0000u);
        csum[2] += __uint_as_float(v[1] << 16); csum[3] += __uint_as_float(v[1] & 0xffff0000u);
      }
    }
    // tile it+1 must have landed (this wave's share) before the barrier publishes it; tile it+2 may stay in flight
    if (it + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  }

  float* sC = reinterpret_cast<float*>(smem2);
#ifdef GEMM_ABLATE
  if (dbg & 1) {
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
    if (sacc == 123.456f) ((float*)p.C)[0] = sacc;
    return;
  }
#endif
  if (TA && do_colsum) {  // combine the 8 k-groups (waves) through LDS: 8 x 256 floats
#pragma unroll
    for (int e = 0; e < 4; ++e) sC[wave * BM2 + lane * 4 + e] = csum[e];
    __syncthreads();
    if (threadIdx.x < BM2) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += sC[w * BM2 + threadIdx.x];
      const int m = m0 + threadIdx.x;
      if (m < p.M) atomicAdd(p.colsum_out + z0 * p.colsum_stride + m, v);
    }
    __syncthreads();
  }

  // ---- epilogue through LDS: the whole 256x128 f32 tile (132 KiB of the 144 KiB) in one pass, then every thread owns 8
  // consecutive columns of a row (16/32-B global accesses).
  constexpr int LDS_C = BN + 4;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row_l = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        sC[row_l * LDS_C + wn * 64 + j * 32 + lr] = acc[i][j][r];
      }
  __syncthreads();
  if (p.atomic) {
    // split-K accumulation.  The plain case (no bias / scale / dropout / row map: every weight gradient) is a bare
    // atomicAdd per element -- the generic per-element epilogue costs ~4x the instructions of the add itself
    const bool plain = p.epi == EPI_STORE && !p.bias && p.alpha == 1.f && p.drop.threshold == 0u && !p.r_on;
    if (plain) {
      float* Cb = (float*)p.C + coff;
#pragma unroll 4
      for (int e = threadIdx.x; e < BM2 * BN; e += 512) {
        const int row_l = e >> 7, col = e & (BN - 1);
        const int m = m0 + row_l, n = n0 + col;
        if (m < p.M && n < p.N) atomicAdd(Cb + (long long)m * p.ldc + (long long)n * p.csc, sC[row_l * LDS_C + col]);
      }
    } else {
      for (int e = threadIdx.x; e < BM2 * BN; e += 512) {
        const int row_l = e >> 7, col = e & (BN - 1);
        const int m = m0 + row_l, n = n0 + col;
        if (m < p.M && n < p.N) epilogue(p, z, coff, m, n, sC[row_l * LDS_C + col]);
      }
    }
    return;
  }
  if ((p.vec_ok & 1) && !(p.N & 3)) {
    // N = 4 (mod 8): the dropout runs of every other row start unaligned; a tile the matrix ends in: per-thread column guard
    // (both only in the instances that need them: the common case keeps its instruction stream)
    if (n0 + BN <= p.N) {
      if (p.N & 7) fast_epilogue_any<8, 32, BN, true>(p, sC, z, coff, m0 + (threadIdx.x >> 4), n0, threadIdx.x >> 4);
      else fast_epilogue_any<8, 32>(p, sC, z, coff, m0 + (threadIdx.x >> 4), n0, threadIdx.x >> 4);
      return;
    }
    if (p.vec_ok & 2) {
      fast_epilogue_any<8, 32, BN, true, true>(p, sC, z, coff, m0 + (threadIdx.x >> 4), n0, threadIdx.x >> 4);
      return;
    }
  }
#pragma unroll 2
  for (int it = 0; it < 8; ++it) {
    const int row_l = (threadIdx.x >> 4) + 32 * it;
    const int c8 = (threadIdx.x & 15) * 8;
    const int m = m0 + row_l, n = n0 + c8;
    if (m < p.M && n < p.N) {
      float v[8];
      const float4 a = *reinterpret_cast<const float4*>(sC + row_l * LDS_C + c8);
      const float4 b = *reinterpret_cast<const float4*>(sC + row_l * LDS_C + c8 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      epilogue_chunk(p, z, coff, m, n, v);
    }
  }
}

template <bool TA, bool TB, bool REG = false>
__global__ __launch_bounds__(512) void gemm_bf16_v2_kernel(GemmP p) {
  drop_resolve(p.drop);
  extern __shared__ __attribute__((aligned(16))) bf16_t smem2[];  // 3 stages x 48 KiB
  gemm_v2_body<TA, TB, REG>(p, smem2, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ---- grouped weight gradients: up to GRP_MAX independent TN problems (same K = token count, f32 atomic accumulation)
// in ONE launch.  A layer's eight weight gradients are 8-32 output tiles each: launched one by one they need split-K 8-16
// to fill the chip (8-16 atomic passes over every gradient, half-empty launches for the 512x512 ones); together they are
// 184 tiles, so split-K 4 fills three rounds of the 256 CUs with twice as long K loops and half the atomic traffic.
#define GRP_MAX 40
struct GroupP {
  const void* A[GRP_MAX]; const void* B[GRP_MAX]; void* C[GRP_MAX]; float* colsum[GRP_MAX];
  int M[GRP_MAX], N[GRP_MAX];
  long long lda[GRP_MAX], ldb[GRP_MAX], ldc[GRP_MAX];
  int tile_begin[GRP_MAX + 1];  // prefix sums of the problems' 256x128 tile counts
  int n, K, splitk, ktiles_per_split;
};
__global__ __launch_bounds__(512) void gemm_bf16_grouped_tn_kernel(GroupP g) {
  extern __shared__ __attribute__((aligned(16))) bf16_t smem2[];
  int pi = 0;
  // The launch read 1.36 GB for ~0.5 GB of unique
  // operands because the tiles of one (problem, K slice) -- which share both operand panels -- are dealt round-robin over
  // the 8 XCDs (8 private L2s).  Here the (problem, K slice) groups are laid out one after the other and every XCD takes a
  // contiguous eighth of that sequence: the tiles running together on an XCD belong to the same group.
  const int tiles_total = g.tile_begin[g.n];
  const int W = tiles_total * g.splitk;
  const int L = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;  // dispatch order: XCD = L % 8
  const int q8w = W >> 3, r8w = W & 7, xw = L & 7;                   // bijection L -> w for any W (XCD x owns q8w or q8w+1 items)
  const int w = (xw < r8w ? xw * (q8w + 1) : r8w * (q8w + 1) + (xw - r8w) * q8w) + (L >> 3);
  while (pi + 1 < g.n && w >= g.splitk * g.tile_begin[pi + 1]) ++pi;  // uniform
  const int ntile_p = g.tile_begin[pi + 1] - g.tile_begin[pi];
  const int rel = w - g.splitk * g.tile_begin[pi];
  const int exp_ks = rel / ntile_p, exp_tile = rel - exp_ks * ntile_p;
  GemmP p;
  p.A = g.A[pi]; p.B = g.B[pi]; p.C = g.C[pi];
  p.M = g.M[pi]; p.N = g.N[pi]; p.K = g.K;
  p.lda = g.lda[pi]; p.ldb = g.ldb[pi]; p.ldc = g.ldc[pi]; p.csc = 1;
  p.transA = 1; p.transB = 1; p.batch = 1; p.nb0 = 1;
  p.sA0 = p.sA1 = p.sB0 = p.sB1 = p.sC0 = p.sC1 = 0;
  p.bias = nullptr; p.alpha = 1.f; p.epi = EPI_STORE; p.c_dt = MI_DT_F32; p.atomic = 1; p.swish_g = 0;
  p.aux_in = nullptr; p.auxin_dt = 0; p.aux_out = nullptr; p.auxout_dt = 0; p.ldaux = 0;
  p.drop.key = 0u; p.drop.threshold = 0u; p.drop.scale = 1.f; p.drop.step = nullptr;
  p.row_len = nullptr; p.rows_per_b = 1; p.rows_inner = 1;
  p.splitk = g.splitk; p.ktiles_per_split = g.ktiles_per_split;
  p.colsum_stride = 0; p.colsum_out = g.colsum[pi];
  p.vec_ok = 0; p.g_on = 0; p.r_on = 0;
  gemm_v2_body<true, true>(p, smem2, exp_tile, exp_ks, 0, /*direct=*/true);
}

// =================================================================================================
// Third structure, for problems with enough 256x256 tiles (N >= 1024 outputs, conv2, split-K wgrad): 256x256x64 block
// tile, 8 waves as 2(M) x 4(N), each 128x64 (8 accumulators).  The 256x128 structure above is bound by its LDS traffic --
// measured: 48 KiB of LDS-DMA landings (64 B/clk) + 128 KiB of fragment reads per K-tile against 1024 MFMA cycles --
// and this tile moves 2/3 of the DMA bytes and 3/4 of the fragment bytes per MFMA.  Two 64-KiB LDS stages (128 KiB): the
// DMA of K-tile it+1 is issued before the MFMAs of tile it and drained (vmcnt(0)) at the barrier that ends the step.
// Same LDS images / swizzles / transpose reads as above; epilogue through a [64][260] f32 window in four rounds.
// =================================================================================================
#define BN4 256
#define NT4_STAGE ((BM2 + BN4) * BK)  // 64 KiB

// one epilogue round of a 256x256 tile: window row rl (0..63) is tile row rbase + (rl >> 5) * hstride + (rl & 31)
// (third / sixth structure: rbase = i * 32, hstride = 128; eighth: rbase = h * 128 + pr * 32, hstride = 64)
__device__ __forceinline__ void v4_round_out(const GemmP& p, const float* sC, int z, long long coff, int m0, int n0, int rbase,
                                          bool fast, const int hstride = 128) {
  constexpr int LDS_C = BN4 + 4;
  if (fast) {
    // 512 threads = 16 rows x 32 column chunks per pass
    const int rl0 = threadIdx.x >> 5;
    fast_epilogue_any<2, 16, BN4>(p, sC, z, coff, m0 + rbase + rl0, n0, rl0);
    fast_epilogue_any<2, 16, BN4>(p, sC + 32 * LDS_C, z, coff, m0 + hstride + rbase + rl0, n0, rl0);
  } else if (p.atomic) {
    const bool plain = p.epi == EPI_STORE && !p.bias && p.alpha == 1.f && p.drop.threshold == 0u && !p.r_on;
    float* Cb = (float*)p.C + coff;
#pragma unroll 4
    for (int e = threadIdx.x; e < 64 * BN4; e += 512) {
      const int rl = e >> 8, col = e & (BN4 - 1);
      const int m = m0 + (rl >> 5) * hstride + rbase + (rl & 31), n = n0 + col;
      if (m < p.M && n < p.N) {
        if (plain) atomicAdd(Cb + (long long)m * p.ldc + (long long)n * p.csc, sC[rl * LDS_C + col]);
        else epilogue(p, z, coff, m, n, sC[rl * LDS_C + col]);
      }
    }
  } else {
    for (int it = 0; it < 4; ++it) {
      const int rl = (threadIdx.x >> 5) + 16 * it;
      const int c8 = (threadIdx.x & 31) * 8;
      const int m = m0 + (rl >> 5) * hstride + rbase + (rl & 31), n = n0 + c8;
      if (m < p.M && n < p.N) {
        float v[8];
        const float4 a = *reinterpret_cast<const float4*>(sC + rl * LDS_C + c8);
        const float4 b = *reinterpret_cast<const float4*>(sC + rl * LDS_C + c8 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        epilogue_chunk(p, z, coff, m, n, v);
      }
    }
  }
}

// ---- epilogue of a 256x256 tile held as acc[4][2] per wave (2 x 4 waves of 128 x 64): four rounds through a [64][BN4+4] f32
// window (round i = the i-th 32-row block of every wave).  Shared by the LDS-DMA structure and the register-prefetch structure.
__device__ __forceinline__ void v4_tile_epilogue(const GemmP& p, f32x16 (&acc)[4][2], float* sC, int z, long long coff, int m0,
                                                 int n0, int wm, int wn, int lr, int lh) {
  constexpr int LDS_C = BN4 + 4;
  const bool fast = (p.vec_ok & 1) && !(p.N & 7) && n0 + BN4 <= p.N && !p.atomic;
  // the rounds are a real loop (the epilogue code exists once); the round's two accumulators are selected by a uniform
  // switch so that acc[][] is never indexed dynamically (that would put all 128 accumulator registers in scratch memory)
#pragma nounroll
  for (int i = 0; i < 4; ++i) {
    f32x16 t0, t1;
    switch (i) {
      case 0: t0 = acc[0][0]; t1 = acc[0][1]; break;
      case 1: t0 = acc[1][0]; t1 = acc[1][1]; break;
      case 2: t0 = acc[2][0]; t1 = acc[2][1]; break;
      default: t0 = acc[3][0]; t1 = acc[3][1]; break;
    }
    if (i) __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row_l = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      sC[row_l * LDS_C + wn * 64 + lr] = t0[r];
      sC[row_l * LDS_C + wn * 64 + 32 + lr] = t1[r];
    }
    __syncthreads();
    v4_round_out(p, sC, z, coff, m0, n0, i * 32, fast);
  }
}

// G: 0 = dense operands, 1 = gathered A rows (conv forward / dgrad), 2 = gathered reduction-major B (conv weight gradient):
// compile-time, so that each instantiation carries only its own DMA state (128 accumulator registers leave little room).
template <bool TA, bool TB, int G>
__global__ __launch_bounds__(512) void gemm_bf16_v4_kernel(GemmP p) {
  drop_resolve(p.drop);
  extern __shared__ __attribute__((aligned(16))) bf16_t smem4[];  // 2 stages x 64 KiB
  const int tn = (p.N + BN4 - 1) / BN4, tm = (p.M + BM2 - 1) / BM2;
  const int ntiles = tm * tn;
  int logical, z, ks;
  if constexpr (G == 2) {
    // conv weight gradient (9 taps x 4 tiles x split-K): the workgroups of ONE K slice -- all taps, all tiles -- share the
    // slice's rows of dY and its (overlapping, tap-shifted) rows of the activation grid, but the plain (x, y, z) order deals
    // them to the 8 XCDs (8 private L2s) so that the workgroups resident on an XCD together belong to different K slices
    // and share nothing (measured: 3.9 GB fetched for 1.6 GB of operands).  Here the (K slice, tap, tile) list is laid out
    // slice-major and every XCD takes a contiguous eighth of it: co-resident workgroups work on the same slice.
    const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
    const int Wt = gx * gy * gz;
    const int L = (int)blockIdx.x + gx * ((int)blockIdx.y + gy * (int)blockIdx.z);  // dispatch order: XCD = L % 8
    const int q8w = Wt >> 3, r8w = Wt & 7, xw = L & 7;
    const int w = (xw < r8w ? xw * (q8w + 1) : r8w * (q8w + 1) + (xw - r8w) * q8w) + (L >> 3);
    const int ncombo = gx * gz;
    ks = w / ncombo;
    const int combo = w - ks * ncombo;
    z = combo / gx;
    logical = combo - z * gx;
  } else {
    const int bid = blockIdx.x;
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    z = blockIdx.z;
    ks = blockIdx.y;
  }
  const int tile_m = logical / tn, tile_n = logical - tile_m * tn;
  const int m0 = tile_m * BM2, n0 = tile_n * BN4;
  const int z0 = z % p.nb0, z1 = z / p.nb0;
  const bf16_t* A = (const bf16_t*)p.A + z0 * p.sA0 + z1 * p.sA1;
  const bf16_t* B = (const bf16_t*)p.B + z0 * p.sB0 + z1 * p.sB1;
  const long long coff = z0 * p.sC0 + z1 * p.sC1;
  if constexpr (!TA && !TB) {
    // EPI_RELU_MASK (the conv stack of the sub-sampling, rows = (utterance, time, ...)): a tile whose rows all lie beyond their
    // utterance's length is zero whatever the product says -- skip the K loop (round 5, SURVEY 8 f1: with unshaped batches 40 % of
    // the tiles of conv2 are such tails).  One utterance per tile only (a tile that crosses into the next utterance is computed).
    // EPI_MUL_POS with row_len (the conv stack's input gradient, round 5): the caller promises that the gate aux_in is <= 0 on every
    // row beyond its utterance's length (conv1's output is masked there), so such rows are zero too -- written through the row map.
    const bool mask_tile = p.epi == EPI_RELU_MASK && !p.r_on;
    const bool gate_tile = p.epi == EPI_MUL_POS && p.row_len != nullptr;
    if ((mask_tile || gate_tile) && p.splitk <= 1 && !p.aux_out && p.c_dt == MI_DT_BF16 && p.csc == 1 && (p.vec_ok & 1) && !(p.N & 7)) {
      const int mlast = min(m0 + BM2, p.M) - 1;
      const int b0 = m0 / p.rows_per_b, b1 = mlast / p.rows_per_b;
      const int t0 = (m0 - b0 * p.rows_per_b) / p.rows_inner;
      if (b0 == b1 && (long long)t0 >= p.row_len[b0]) {
        bf16_t* Cz = (bf16_t*)p.C + coff;
        const int ncol = min(BN4, p.N - n0), nrow = mlast - m0 + 1, cpr = ncol >> 3;  // (vec_ok: N % 8 == 0, rows 16-byte aligned)
        const u32x4 zero = {0u, 0u, 0u, 0u};
        for (int e = threadIdx.x; e < nrow * cpr; e += 512) {
          const int r = e / cpr, c = e - r * cpr;
          *reinterpret_cast<u32x4*>(Cz + crow(p, m0 + r) * p.ldc + n0 + c * 8) = zero;
        }
        return;
      }
    }
  }
  const int nk_total = (p.K + BK - 1) / BK;
  int kt0 = 0, kt1 = nk_total;
  if (p.splitk > 1) {
    kt0 = ks * p.ktiles_per_split;
    kt1 = min(nk_total, kt0 + p.ktiles_per_split);
    if (kt0 >= kt1) return;
  }
  const int nk = kt1 - kt0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane & 31, lh = lane >> 5;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  DmaSrc<BM2> dA;
  DmaSrc<BN4> dB;
  DmaGather<BM2> gA;
  constexpr bool gatherA = !TA && G == 1;
  if constexpr (gatherA) gather_setup<BM2>(gA, p, A, m0);
  else dma_setup<TA, BM2>(dA, A, p.lda, m0, p.M, kt0 * BK);
  DmaGatherT<BN4> gB;
  constexpr bool gatherB = TB && G == 2;
  if constexpr (gatherB) gatherT_setup<BN4>(gB, p, n0, p.N, kt0 * BK);
  else dma_setup<TB, BN4>(dB, B, p.ldb, n0, p.N, kt0 * BK);
  const bool ktail = (p.K & (BK - 1)) != 0;
  // conv weight gradient with row_len (round 5): K runs over the output positions (b, i, j); a K-tile whose 64
  // positions all lie beyond their utterance's length multiplies zeros (dY is masked there) -- neither loaded nor multiplied.
  // One utterance per tile only (a tile that crosses into the next utterance is computed).
  // Called once per tile in increasing order: the utterance of the tile's first position is tracked incrementally, its length is
  // re-read only when the tile sequence enters the next utterance (a load per K-tile would sit in front of every DMA issue).
  int u_b = 0, u_row0 = 0, u_valid = 0;   // utterance of the current tile, its first position, its number of valid positions
  if constexpr (gatherB) {
    if (p.row_len) {
      u_b = (kt0 * BK) / p.rows_per_b;
      u_row0 = u_b * p.rows_per_b;
      u_valid = (int)min((long long)p.rows_per_b, p.row_len[u_b] * p.rows_inner);
    }
  }
  auto dead = [&](int it) -> bool {   // uniform
    if constexpr (!gatherB) return false;
    if (!p.row_len) return false;
    const int k0 = (kt0 + it) * BK, k1 = min(k0 + BK, p.K) - 1;
    while (k0 >= u_row0 + p.rows_per_b) {
      ++u_b; u_row0 += p.rows_per_b;
      u_valid = (int)min((long long)p.rows_per_b, p.row_len[u_b] * p.rows_inner);
    }
    return k1 < u_row0 + p.rows_per_b && k0 - u_row0 >= u_valid;
  };
  auto issue_a = [&](int it) {
    bf16_t* st = smem4 + (it & 1) * NT4_STAGE;
    if constexpr (gatherA) gather_issue<BM2>(gA, p, kt0 + it, st);
    else dma_issue<BM2>(dA, it, (kt0 + it) * BK, p.K, ktail, st);
  };
  auto issue_b = [&](int it, bool load) {
    bf16_t* st = smem4 + (it & 1) * NT4_STAGE;
    if constexpr (gatherB) gatherT_issue<BN4>(gB, p, (const bf16_t*)p.B, z0, st + BM2 * BK, load);
    else dma_issue<BN4>(dB, it, (kt0 + it) * BK, p.K, ktail, st + BM2 * BK);
  };
  auto issue = [&](int it) -> bool {   // returns whether tile `it` carries work
    const bool live = !dead(it);
    if (live) issue_a(it);
    issue_b(it, live);
    return live;
  };
  const int cs_step = tn < 8 ? tn : 8;  // column-sum rows dealt round-robin to the workgroups sharing this A tile
  const bool do_colsum = TA && p.colsum_out != nullptr && tile_n < cs_step;
  float csum[4] = {0.f, 0.f, 0.f, 0.f};

  bool live_cur = issue(0), live_next = false;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  if constexpr (TA && TB) {
  // both operands reduction-major: transpose reads by inline asm (see tr_rd), one k-step ahead of the MFMAs
  FragU fa[2][4], fb[2][2];
  uint32_t abase[4], bbase[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) abase[i] = tr_base<BM2>(smem4, wm * 128 + i * 32, lane);
#pragma unroll
  for (int j = 0; j < 2; ++j) bbase[j] = tr_base<BN4>(smem4 + BM2 * BK, wn * 64 + j * 32, lane);
#define V4_RD(S, KK)                                                                              \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) tr_frag_rd<BN4, KK>(fb[S][j], bbase[j] + sb);     \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) tr_frag_rd<BM2, KK>(fa[S][i], abase[i] + sb)
#define V4_WAIT(N, S)                                                                                          \
  asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                     \
               : "+v"(fa[S][0].v), "+v"(fa[S][1].v), "+v"(fa[S][2].v), "+v"(fa[S][3].v), "+v"(fb[S][0].v),     \
                 "+v"(fb[S][1].v)                                                                              \
               :                                                                                               \
               : "memory");                                                                                    \
  __builtin_amdgcn_sched_barrier(0)
#define V4_MM(S)                                                                                               \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                  \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[S][i].v, fb[S][j].v, acc[i][j], 0, 0, 0);         \
  __builtin_amdgcn_sched_barrier(0)
  for (int it = 0; it < nk; ++it) {
    // tile it+1 goes to the other stage: every wave passed the barrier after its last read of it
    const uint32_t sb = (uint32_t)((it & 1) * (NT4_STAGE * 2));
    if (it + 1 < nk) live_next = issue(it + 1);
    __builtin_amdgcn_sched_barrier(0);
    if (live_cur) {
    V4_RD(0, 0);
    V4_RD(1, 1); V4_WAIT(12, 0); V4_MM(0);
    V4_RD(0, 2); V4_WAIT(12, 1); V4_MM(1);
    V4_RD(1, 3); V4_WAIT(12, 0); V4_MM(0);
    V4_WAIT(0, 1); V4_MM(1);
    }
    if (do_colsum && live_cur) {  // thread -> (k-group of 8 rows = wave, 4 consecutive columns = lane)
      const int col = lane * 4;
      const uint32_t a_addr = lds_addr(smem4) + sb;
      for (int kr = tile_n; kr < 8; kr += cs_step) {
        const int krow = wave * 8 + kr;
        const int off = krow * BM2 + (((col >> 3) ^ ((krow & 3) << 2)) << 3) + (col & 7);
        const u32x2 v = lds_rd64_sync(a_addr + (uint32_t)(off * 2));
        csum[0] += __uint_as_float(v[0] << 16); csum[1] += __uint_as_float(v[0] & 0xffff0000u);
        csum[2] += __uint_as_float(v[1] << 16); csum[3] += __uint_as_float(v[1] & 0xffff0000u);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    live_cur = live_next;
  }
#undef V4_RD
#undef V4_WAIT
#undef V4_MM
  } else {
  for (int it = 0; it < nk; ++it) {
    if (it + 1 < nk) (void)issue(it + 1);  // other stage: every wave passed the barrier after its last read of it
    const bf16_t* a_s = smem4 + (it & 1) * NT4_STAGE;
    const bf16_t* b_s = a_s + BM2 * BK;
    // (measured: pipelining these reads one k-step ahead with inline-asm counted waits -- 12 reads in flight under the 8 MFMAs
    //  of the previous k-step -- changes nothing, 1060 vs 1070 TFLOP/s at 8192^3: the loop is not bound by fragment-read
    //  latency.  profiles/r2_gemm_structures.md)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 af[4], bfr[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = frag_v2<TB, BN4>(b_s, wn * 64 + j * 32, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = frag_v2<TA, BM2>(a_s, wm * 128 + i * 32, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (TA && do_colsum) {  // thread -> (k-group of 8 rows = wave, 4 consecutive columns = lane)
      const int col = lane * 4;
      for (int kr = tile_n; kr < 8; kr += cs_step) {
        const int krow = wave * 8 + kr;
        const int off = krow * BM2 + (((col >> 3) ^ ((krow & 3) << 2)) << 3) + (col & 7);
        const u32x2 v = *reinterpret_cast<const u32x2*>(a_s + off);
        csum[0] += __uint_as_float(v[0] << 16); csum[1] += __uint_as_float(v[0] & 0xffff0000u);
        csum[2] += __uint_as_float(v[1] << 16); csum[3] += __uint_as_float(v[1] & 0xffff0000u);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  }

  float* sC = reinterpret_cast<float*>(smem4);
  if (TA && do_colsum) {
#pragma unroll
    for (int e = 0; e < 4; ++e) sC[wave * BM2 + lane * 4 + e] = csum[e];
    __syncthreads();
    if (threadIdx.x < BM2) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += sC[w * BM2 + threadIdx.x];
      const int m = m0 + threadIdx.x;
      if (m < p.M) atomicAdd(p.colsum_out + z0 * p.colsum_stride + m, v);
    }
    __syncthreads();
  }

  v4_tile_epilogue(p, acc, reinterpret_cast<float*>(smem4), z, coff, m0, n0, wm, wn, lr, lh);
}

// =================================================================================================
// Sixth structure: the 256x256x64 tile of the third structure with its operands PREFETCHED THROUGH REGISTERS, two K-tiles deep.
// Why: the LDS-DMA loop above can keep ONE K-tile in flight (two 64-KiB stages are all the LDS holds) and drains it at the
// barrier that ends the step, so every K-tile pays one full memory round trip: 2.0-2.4 us per step measured (8.4 MFLOP
// per step and CU = 1.07 PFLOP/s on the chip at best, 0.88 at K = 512) against 0.86 us of MFMA work -- the loop is bound
// by the latency of one transfer, not by LDS bandwidth or the matrix pipe (the ablations of profiles/r2_gemm_structures.md
// fit: without MFMAs the step still takes 1.8 us; a 4-wave / 128x128-per-wave build with 1/3 less LDS traffic ran at the
// same rate).  Bytes in flight are what hides latency, and registers are where a CU has room for them: every thread holds
// the 8 + 8 16-byte chunks of TWO future K-tiles (64 VGPRs), loaded with ordinary global loads three steps before the MFMAs
// that use them, and copies one tile per step into the free LDS stage with ds_write_b128 (the same [rows][64] images and
// chunk swizzle as the DMA path, so the fragment reads are unchanged).  128 KiB are in flight per CU instead of <= 64.
// The compiler's own counted vmcnt does the pipelining: no LDS-DMA, hence no vmcnt(0) in front of LDS reads.
// K-contiguous operands (NT), K % 128 == 0, K >= 256; everything else stays on the third structure.
// =================================================================================================
struct RegTile { u32x4 a[4]; u32x4 b[4]; };  // this thread's chunks of one K-tile: A rows (tid + i*512) >> 3, B likewise

template <int G>
__global__ __launch_bounds__(512) void gemm_bf16_v6_kernel(GemmP p) {
  drop_resolve(p.drop);
  extern __shared__ __attribute__((aligned(16))) bf16_t smem6[];  // 2 stages x 64 KiB
  const int tn = (p.N + BN4 - 1) / BN4, tm = (p.M + BM2 - 1) / BM2;
  const int ntiles = tm * tn;
  const int bid = blockIdx.x;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int z = blockIdx.z;
  const int tile_m = logical / tn, tile_n = logical - tile_m * tn;
  const int m0 = tile_m * BM2, n0 = tile_n * BN4;
  const int z0 = z % p.nb0, z1 = z / p.nb0;
  const char* A = (const char*)((const bf16_t*)p.A + z0 * p.sA0 + z1 * p.sA1);
  const char* B = (const char*)((const bf16_t*)p.B + z0 * p.sB0 + z1 * p.sB1);
  const long long coff = z0 * p.sC0 + z1 * p.sC1;
  const int nk = p.K / BK;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane & 31, lh = lane >> 5;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // per-thread byte offsets of its chunks at K-tile 0 (rows past the matrix are clamped: their products are never stored);
  // the K-tile advances all of them by the uniform BK * 2 bytes
  uint32_t oa[4], ob[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = threadIdx.x + i * 512;
    const int r = q >> 3, ck = q & 7;
    const int gck = ck ^ ((r >> 1) & 7);
    int ga = m0 + r; ga = ga < p.M ? ga : p.M - 1;
    int gb = n0 + r; gb = gb < p.N ? gb : p.N - 1;
    oa[i] = (uint32_t)((long long)ga * p.lda + gck * 8) * 2u;
    ob[i] = (uint32_t)((long long)gb * p.ldb + gck * 8) * 2u;
  }
  // (a K-tile index past the end reads the 16-byte zero page with all offsets masked to 0: the loop body has no branch, so the
  //  compiler's wait-count model of the two register sets stays exact -- with `if (it + 3 < nk)` around the loads it merged the
  //  paths and drained vmcnt(0) in front of every ds_write, i.e. the second tile in flight was lost)
  auto load_tile = [&](RegTile& t, int it) {
    const bool in = it < nk;
    const uint32_t msk = in ? 0xffffffffu : 0u;
    const char* a = in ? A + (long long)it * (BK * 2) : reinterpret_cast<const char*>(g_zero16);
    const char* b = in ? B + (long long)it * (BK * 2) : reinterpret_cast<const char*>(g_zero16);
#pragma unroll
    for (int i = 0; i < 4; ++i) t.a[i] = *reinterpret_cast<const u32x4*>(a + (oa[i] & msk));
#pragma unroll
    for (int i = 0; i < 4; ++i) t.b[i] = *reinterpret_cast<const u32x4*>(b + (ob[i] & msk));
  };
  auto store_tile = [&](const RegTile& t, int stage) {
    bf16_t* st = smem6 + stage * NT4_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(st + (threadIdx.x + i * 512) * 8) = t.a[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(st + BM2 * BK + (threadIdx.x + i * 512) * 8) = t.b[i];
  };
  auto compute = [&](int stage) {
    const bf16_t* a_s = smem6 + stage * NT4_STAGE;
    const bf16_t* b_s = a_s + BM2 * BK;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 af[4], bfr[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = frag_v2<false, BN4>(b_s, wn * 64 + j * 32, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = frag_v2<false, BM2>(a_s, wm * 128 + i * 32, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };

  // one step with the copy of the NEXT tile (register set t -> stage `wr`) and the loads of the tile after the next two spread
  // over the four k-steps of the CURRENT tile's MFMAs: a ds_write_b128 occupies the LDS store path for ~13 cycles per wave
  // (8 waves x 8 stores = ~830 of the step's 2048 MFMA cycles) -- left to the compiler they all sink to the end of the step,
  // where only the barrier is left to hide them (G == 0: compiler order, G == 1: this order; A/B in profiles/r3_gemm_structures.md)
  auto step_spread = [&](RegTile& t, int wr, int rd, int next_it) {
    const bf16_t* a_s = smem6 + rd * NT4_STAGE;
    const bf16_t* b_s = a_s + BM2 * BK;
    bf16_t* st = smem6 + wr * NT4_STAGE;
    const bool in = next_it < nk;
    const uint32_t msk = in ? 0xffffffffu : 0u;
    const char* ga = in ? A + (long long)next_it * (BK * 2) : reinterpret_cast<const char*>(g_zero16);
    const char* gb = in ? B + (long long)next_it * (BK * 2) : reinterpret_cast<const char*>(g_zero16);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 af[4], bfr[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = frag_v2<false, BN4>(b_s, wn * 64 + j * 32, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = frag_v2<false, BM2>(a_s, wm * 128 + i * 32, kk, lane);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int i = (kk & 1) * 2 + c;
        if (kk < 2) {
          *reinterpret_cast<u32x4*>(st + (threadIdx.x + i * 512) * 8) = t.a[i];
          t.a[i] = *reinterpret_cast<const u32x4*>(ga + (oa[i] & msk));
        } else {
          *reinterpret_cast<u32x4*>(st + BM2 * BK + (threadIdx.x + i * 512) * 8) = t.b[i];
          t.b[i] = *reinterpret_cast<const u32x4*>(gb + (ob[i] & msk));
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  RegTile r0, r1;
  load_tile(r0, 0);
  load_tile(r1, 1);
  store_tile(r0, 0);
  load_tile(r0, 2);
  __syncthreads();
  // step `it` (unrolled by two so that the register sets alternate statically; nk is even): tile it+1 (loaded two steps ago)
  // goes from its registers into the stage that step it-1 has finished reading, the registers take tile it+3, then the MFMAs
  // of tile it.  (The copy of a tile past the end into the idle stage is harmless: nothing reads it.)
  for (int it = 0; it < nk; it += 2) {
    if constexpr (G == 1) {
      step_spread(r1, 1, 0, it + 3);
      __syncthreads();
      step_spread(r0, 0, 1, it + 4);
      __syncthreads();
    } else {
      store_tile(r1, 1);
      load_tile(r1, it + 3);
      compute(0);
      __syncthreads();
      store_tile(r0, 0);
      load_tile(r0, it + 4);
      compute(1);
      __syncthreads();
    }
  }
  v4_tile_epilogue(p, acc, reinterpret_cast<float*>(smem6), z, coff, m0, n0, wm, wn, lr, lh);
}

// =================================================================================================
// Eighth structure: the 256x256x64 tile on v_mfma_f32_16x16x32_bf16 with a PHASE-STAGGERED K loop (round 6).
// The 256x256 structures above run all eight waves in lock step: every wave reads its fragments, then every wave multiplies --
// the two waves of a SIMD take turns at ONE matrix pipe and meet at the same barrier with nothing to overlap (PMC, round 5: 2-2.5x
// the wave cycles of a 4-wave kernel for the same MFMA cycles, LDS issue stalls 2-3x).  Here the two wave rows (waves 0-3 / 4-7 =
// one wave of each SIMD) run ONE BARRIER APART: while one wave of a SIMD multiplies (16 MFMAs), its partner issues the LDS reads
// and LDS-DMA of its next phase.  A K-tile is four phases; per phase and wave:
//     fragment reads of the quadrant's new operand half (12 / 4 / 8 / 0 x 16 B)  |  2 x global_load_lds (one 16-KiB half-tile, 7
//     half-tiles ahead)  |  barrier  |  lgkmcnt(0)  |  16 MFMAs = one 64x32 quadrant of the wave's 128x64 block x K = 64  |  barrier
// LDS: two K-tile buffers of four half-tile images (A rows 0-127 | 128-255 | B rows 0-127 | 128-255; 16 KiB each) = 128 KiB.
// Wave (wm, wn) owns rows wm*64..+63 of EACH A half and columns wn*32..+31 of EACH B half, so a phase needs exactly one new
// half-tile:  p0 (A0,B0)  p1 (A0,B1)  p2 (A1,B1)  p3 (A1,B0).
// Half-tile images, one per operand layout (TN = both operands reduction-major, the weight gradients):
//   * K-contiguous ([rows][K]): image [128 rows][64 k], chunk swizzle c ^= (row >> 1) & 7 on the DMA SOURCE, ds_read_b128: lane l
//     reads row l & 15, chunk 4 ks + (l >> 4) -- conflict-free in the instruction's four 16-lane groups;
//   * reduction-major ([K][rows]): image [64 k][128 rows] as in memory, chunk swizzle c ^= ((k & 3) << 2) ^ (((k >> 3) & 1) << 1),
//     two ds_read_b64_tr_b16 per fragment (lane t of a 16-lane group supplies &img[k0 + t/4][row0 + (t%4)*4] and receives
//     img[k0..k0+3][row0 + t]); the second swizzle term keeps the two 16-lane groups of a 32-lane half (k0 and k0 + 8) off each
//     other's banks.
// Pipeline bookkeeping (vmcnt counts this wave's LDS-DMA instructions, 2 per half-tile; half-tiles are staged in the order
// B0, A0, B1, A1 of tile t, t+1, ...; phase p of tile t stages half-tile index 4t + p + 7):
//   * ONE counted wait per K-tile, in phase 3 before its first barrier: vmcnt(6) leaves the three half-tiles staged in phases
//     1-3 (tile t+2) in flight and retires everything of tile t+1, which is first read one phase -- and, for the other wave row,
//     at least one barrier -- later.  Never vmcnt(0) inside the loop (only for the last-but-one tile, when nothing follows).
//     Every half-tile is staged whatever happens (a chunk outside the matrix, past K, or of a K-tile that is skipped reads the
//     zero page): the count must not depend on the data.
//   * a half-tile buffer is re-staged two phases after its last read, except B0 (read in phase 0, re-staged in phase 1): its
//     reads are issued FIRST in phase 0 and retired by a counted lgkmcnt before that phase's first barrier.
// G: 0 dense, 1 = gathered A rows (conv forward / input gradient, NT), 2 = gathered reduction-major B (conv weight gradient, TN).
// NT needs whole K-tiles; at least two K-tiles per workgroup.  Measured (one box, uniform random operands, profiles/r6_gemm_8phase.md):
// 4096^3 1.34-1.37 PFLOP/s (third structure 1.09-1.10), conv2 forward 997 -> 1220 TFLOP/s; without the stagger 1.16 / 1.10;
// s_setprio around the MFMA block: no effect here (kept off).
// Epilogue: four rounds through the same [64][260] f32 window and round-out as the third structure.
// =================================================================================================
#define V8_HALF_B 16384
#define V8_BUF_B 65536
// -DV8_TRACE (tools/ab_build.py variant, never the shipped build): wave 0 of every workgroup stamps s_memrealtime (100 MHz, chip-wide)
// and s_memtime (shader clock) at entry, after the prologue, after the K loop and at exit: 8 words per workgroup of one launch
#ifdef V8_TRACE
__device__ unsigned long long* g_v8_trace = nullptr;
extern "C" int mi355x_gemm_debug_trace(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_v8_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#define V8_STAMP(i)                                                                                              \
  if (g_v8_trace && threadIdx.x == 0) {                                                                          \
    unsigned long long tr_, tc_;                                                                                 \
    asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_), "=s"(tc_)::"memory");  \
    const int wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                              \
    g_v8_trace[wg_ * 16 + 2 * (i)] = tr_; g_v8_trace[wg_ * 16 + 2 * (i) + 1] = tc_;                              \
  }
#else
#define V8_STAMP(i)
#endif
template <int OFF>
__device__ __forceinline__ void v8_rd128(bf16x8& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

// ---- fast epilogue round of the eighth structure: 64 window rows x 256 columns, window = [64][256] f32 with the 16-byte chunks of
// a row XOR-swizzled by (row & 7) (no padding: two windows fill the 128 KiB, so a round's writes never wait for the previous round's
// readers).  One thread = 8 consecutive columns of FOUR rows (window rows rl0 + 16 it; tile rows m_base + (it >> 1) * hstride + rl0 +
// 16 * (it & 1)).  All aux_in loads, then all eight LDS reads of the thread go out before the first use (the generic fast_epilogue
// walks its rows one after the other: 16 dependent LDS round trips per tile were 4 of the epilogue's 8 us, profiles/r6_gemm_8phase.md);
// the bias is loaded once per tile by the caller.  Same arithmetic in the same order as fast_epilogue / epilogue8 (bit-identical).
template <int EPI>
__device__ __forceinline__ void v8_round_fast(const GemmP& p, const float* win, const float (&b8)[8], int z, long long coff, int m_base,
                                              int hstride, int n0) {
  constexpr bool AUX_IN = EPI == EPI_RESID || EPI == EPI_DSWISH || EPI == EPI_MUL_POS;
  const int k = threadIdx.x & 31, rl0 = threadIdx.x >> 5;
  const int n = n0 + k * 8;
  const bool lin = (EPI != EPI_MUL_POS) || !p.r_on;
  int mrow[4];
  long long ci[4], ai[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    mrow[it] = m_base + (it >> 1) * hstride + rl0 + 16 * (it & 1);
    const long long mr = lin ? (long long)mrow[it] : crow(p, mrow[it] < p.M ? mrow[it] : p.M - 1);
    ci[it] = coff + mr * p.ldc + n;
    ai[it] = coff + mr * p.ldaux + n;
  }
  float aux[AUX_IN ? 4 : 1][8];
  if (AUX_IN) {
#pragma unroll
    for (int it = 0; it < 4; ++it)
      if (mrow[it] < p.M) ld8x(p.aux_in, ai[it], EPI == EPI_RESID ? MI_DT_F32 : p.auxin_dt, aux[it]);
  }
  // lanes 16-31 of each half wave read their high 16 bytes first: every 16-lane service group of ds_read_b128 then covers the 16
  // slots of the 256-byte bank row once (the row's XOR only permutes them)
  const int hi = (threadIdx.x >> 4) & 1, sw = rl0 & 7;
  float4 x[4], y[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const float* row = win + (rl0 + 16 * it) * 256;
    x[it] = *reinterpret_cast<const float4*>(row + (((2 * k + hi) ^ sw) << 2));
    y[it] = *reinterpret_cast<const float4*>(row + (((2 * k + 1 - hi) ^ sw) << 2));
  }
  const uint32_t dbase = (uint32_t)z * (uint32_t)(p.M * p.N) + (uint32_t)n;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int m = mrow[it];
    if (m >= p.M) continue;
    const float4 a = hi ? y[it] : x[it], b = hi ? x[it] : y[it];
    float v[8] = {a.x + b8[0], a.y + b8[1], a.z + b8[2], a.w + b8[3], b.x + b8[4], b.y + b8[5], b.z + b8[6], b.w + b8[7]};
    float dm[8];
    drop_mask8(p.drop, dbase + (uint32_t)m * (uint32_t)p.N, dm);
    if (EPI == EPI_STORE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= p.alpha * dm[j];
    } else if (EPI == EPI_SWISH_DROP) {
      if (p.swish_g) {
        float g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) swish_pair(v[j], dm[j], v[j], g[j]);
        st8x(p.aux_out, ai[it], p.auxout_dt, g);
      } else {
        st8x(p.aux_out, ai[it], p.auxout_dt, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = swishf_(v[j]) * dm[j];
      }
    } else if (EPI == EPI_RESID) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = aux[AUX_IN ? it : 0][j] + p.alpha * v[j] * dm[j];
    } else if (EPI == EPI_DSWISH) {
      if (p.swish_g) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= aux[AUX_IN ? it : 0][j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * dm[j] * swish_grad(aux[AUX_IN ? it : 0][j]);
      }
    } else if (EPI == EPI_RELU_MASK) {
      const int bb = m / p.rows_per_b;
      const int t = (m - bb * p.rows_per_b) / p.rows_inner;
      const bool ok = (long long)t < p.row_len[bb];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (ok && v[j] > 0.f) ? v[j] : 0.f;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = aux[AUX_IN ? it : 0][j] > 0.f ? v[j] : 0.f;
    }
#if defined(V8_EPI_ABL) && V8_EPI_ABL == 3
    asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(ci[it]));
#else
    st8x(p.C, ci[it], p.c_dt, v);
#endif
  }
}

// AH = A halves of the tile: 2 = the 256x256 tile described above; 1 = a 128x256 tile (dense NT only) for problems whose 256x256
// tiles would leave half the chip idle (N = 512 at M = 16032: 126 tiles): one A half, both B halves, two phases per K-tile
// (p0 (A0,B0), p1 (A0,B1)), THREE K-tile buffers of three half-tile images (A0 | B0 | B1, 144 KiB) filled two K-tiles ahead -- a
// K-tile is only 1 024 MFMA cycles here, one tile of lead would not cover the DMA latency.  Phase 0 stages B0 of tile t+2, phase 1
// its B1 and A0 and waits vmcnt(6): all of tile t+1 landed, tile t+2 in flight; every buffer is re-staged two or three phases
// after its last read (no early-retire trick needed).
template <int G, bool TN, int AH = 2>
__device__ __forceinline__ void gemm_v8_body(const GemmP& p, bf16_t* smem8, const int tile_m, const int tile_n, const int kslice,
                                             const int z) {
  const int tn = (p.N + BN4 - 1) / BN4;
  static_assert(AH == 2 || (AH == 1 && G == 0 && !TN), "the one-A-half tile: dense K-contiguous operands");
  constexpr int BMT = 128 * AH;                      // tile rows
  constexpr int BUF_B = (AH + 2) * V8_HALF_B;        // bytes of one K-tile buffer: A halves | B0 | B1
  const int m0 = tile_m * BMT, n0 = tile_n * BN4;
  const int z0 = z % p.nb0, z1 = z / p.nb0;
  const bf16_t* A = (const bf16_t*)p.A + z0 * p.sA0 + z1 * p.sA1;
  const bf16_t* B = (const bf16_t*)p.B + z0 * p.sB0 + z1 * p.sB1;
  const long long coff = z0 * p.sC0 + z1 * p.sC1;
  if constexpr (!TN) {
    // row tiles that lie wholly beyond their utterance (ReLU + time mask forward, gated input gradient): zero-filled without a
    // K loop -- same rule as the third structure
    const bool mask_tile = p.epi == EPI_RELU_MASK && !p.r_on;
    const bool gate_tile = p.epi == EPI_MUL_POS && p.row_len != nullptr;
    if ((mask_tile || gate_tile) && p.splitk <= 1 && !p.aux_out && p.c_dt == MI_DT_BF16 && p.csc == 1 && (p.vec_ok & 1) && !(p.N & 7)) {
      const int mlast = min(m0 + BMT, p.M) - 1;
      const int b0 = m0 / p.rows_per_b, b1 = mlast / p.rows_per_b;
      const int t0 = (m0 - b0 * p.rows_per_b) / p.rows_inner;
      if (b0 == b1 && (long long)t0 >= p.row_len[b0]) {
        bf16_t* Cz = (bf16_t*)p.C + coff;
        const int ncol = min(BN4, p.N - n0), nrow = mlast - m0 + 1, cpr = ncol >> 3;
        const u32x4 zero = {0u, 0u, 0u, 0u};
        for (int e = threadIdx.x; e < nrow * cpr; e += 512) {
          const int r = e / cpr, c = e - r * cpr;
          *reinterpret_cast<u32x4*>(Cz + crow(p, m0 + r) * p.ldc + n0 + c * 8) = zero;
        }
        return;
      }
    }
  }
  const int nk_total = (p.K + BK - 1) / BK;
  int kt0 = 0, kt1 = nk_total;
  if (p.splitk > 1) {
    kt0 = kslice * p.ktiles_per_split;
    kt1 = min(nk_total, kt0 + p.ktiles_per_split);
    if (kt0 >= kt1) return;
  }
  const int nk = kt1 - kt0;  // >= 2 by contract
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  f32x4 acc[AH][4][2][2];  // [A half][m fragment][B half][n fragment]: rows h*128 + wm*64 + mi*16, columns hb*128 + wn*32 + ni*16
#pragma unroll
  for (int h = 0; h < AH; ++h)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int hb = 0; hb < 2; ++hb)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[h][mi][hb][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- LDS-DMA sources: this thread's two 16-byte chunks (i = 0, 1) of each half-tile (h), as 32-bit byte offsets at K-tile kt0;
  // a K-tile advances all of an operand's offsets by one uniform step.  A chunk that must not be read (column outside the matrix,
  // row past K, tap outside the grid, skipped K-tile) takes the zero page instead.
  //   NT: chunk q = tid + 512 i -> row q >> 3, LDS chunk position q & 7 <- source chunk (q & 7) ^ ((row >> 1) & 7); rows past the
  //       matrix are clamped (their products are never stored)
  //   TN: chunk q -> k-row q >> 4, position q & 15 <- source chunk (q & 15) ^ f(k); columns past the matrix read zeros
  //   G == 1: an A row is an output position; offset of the channel vector at its origin src[b][i*si][j*sj][.], bit t of gin = tap t
  //       of that position lies inside the grid; the K-tile's tap / channel offset are uniform and tracked incrementally
  //   G == 2: a B k-row is an output position (shared by both halves); its (i, j) and origin offset advance by 64 positions per tile
  uint32_t oA[2][2], oB[2][2], gin[2][2];
  uint32_t okbits = 0u;  // TN: bit h*2+i = A chunk inside the matrix, bit 4+h*2+i = B chunk
  int g2i[2], g2j[2], g2off[2], g2row[2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = threadIdx.x + i * 512;
      gin[h][i] = 0u;
      if constexpr (!TN) {
        const int r = q >> 3, gck = (q & 7) ^ ((r >> 1) & 7);
        int ga = m0 + h * 128 + r; ga = ga < p.M ? ga : p.M - 1;
        int gb = n0 + h * 128 + r; gb = gb < p.N ? gb : p.N - 1;
        oB[h][i] = (uint32_t)((long long)gb * p.ldb + gck * 8) * 2u;
        if constexpr (G == 1) {
          const int per_b = p.g_nI * p.g_nJ;
          const int b = ga / per_b, rr = ga - b * per_b;
          const int oi = rr / p.g_nJ, oj = rr - oi * p.g_nJ;
          const int gi = oi * p.g_si, gj = oj * p.g_sj;
          oA[h][i] = (uint32_t)((((long long)b * p.g_SI + gi) * p.g_SJ + gj) * p.g_C + gck * 8) * 2u;
          uint32_t in = 0u;
          for (int t = 0; t < p.g_ntaps; ++t) {
            const int di = tap_delta(p.g_dip, t), dj = tap_delta(p.g_djp, t);
            if ((unsigned)(gi + di) < (unsigned)p.g_SI && (unsigned)(gj + dj) < (unsigned)p.g_SJ) in |= 1u << t;
          }
          gin[h][i] = in;
        } else {
          oA[h][i] = (uint32_t)((long long)ga * p.lda + gck * 8) * 2u;
        }
      } else {
        const int k = q >> 4, c = (q & 15) ^ ((k & 3) << 2) ^ (((k >> 3) & 1) << 1);
        const int ga = m0 + h * 128 + c * 8, gb = n0 + h * 128 + c * 8;
        if (ga < p.M) okbits |= 1u << (h * 2 + i);
        if (gb < p.N) okbits |= 16u << (h * 2 + i);
        oA[h][i] = (uint32_t)((long long)k * p.lda + ga) * 2u;
        if constexpr (G == 2) {
          oB[h][i] = (uint32_t)gb * 2u;  // column only: the position's origin is g2off
          if (h == 0) {
            const int m = kt0 * BK + k;
            const int per_b = p.g_nI * p.g_nJ;
            const int b = m / per_b, rr = m - b * per_b;
            g2i[i] = rr / p.g_nJ;
            g2j[i] = rr - g2i[i] * p.g_nJ;
            g2row[i] = m;
            g2off[i] = ((b * p.g_SI + g2i[i] * p.g_si) * p.g_SJ + g2j[i] * p.g_sj) * p.g_C;
          }
        } else {
          oB[h][i] = (uint32_t)((long long)k * p.ldb + gb) * 2u;
        }
      }
    }
  const long long stepA = TN ? (long long)BK * p.lda * 2 : (long long)BK * 2;  // bytes per K-tile
  const long long stepB = TN ? (long long)BK * p.ldb * 2 : (long long)BK * 2;
  const char* Ab = (const char*)A + (G == 1 ? 0ll : kt0 * stepA);
  const char* Bb = (const char*)B + (G == 2 ? 0ll : kt0 * stepB);
  char* const lds8 = (char*)smem8;
  const char* const zpage = reinterpret_cast<const char*>(g_zero16);
  const int krem = p.K - (nk_total - 1) * BK;      // rows of the last K-tile (TN; NT has whole K-tiles)
  const int k_lo = threadIdx.x >> 4;                // TN: k-row of this thread's chunk i is k_lo + 32 i
  // G == 1 cursor: tap / channel offset of the K-tile whose A halves are being staged (A0 of a tile is staged in front of its A1:
  // the cursor advances in front of every A0)
  int g_tap = 0, g_c0 = 0;
  long long g_toff = 0;
  auto gather_offs = [&]() {
    const int di = tap_delta(p.g_dip, g_tap), dj = tap_delta(p.g_djp, g_tap);
    g_toff = (((long long)di * p.g_SJ + dj) * p.g_C + g_c0) * 2;
  };
  auto gather_next = [&]() {
    g_c0 += BK;
    if (g_c0 >= p.g_C) { g_c0 = 0; ++g_tap; }
    gather_offs();
  };
  if constexpr (G == 1) {
    g_tap = (kt0 * BK) / p.g_C;
    g_c0 = kt0 * BK - g_tap * p.g_C;
    gather_offs();
  }
  // G == 2: the tap is the batch index; positions advance after the tile's second B half (B0, B1 of one tile use the same state)
  int g2_toff = 0, g2_di = 0, g2_dj = 0;
  if constexpr (G == 2) {
    g2_di = tap_delta(p.g_dip, z0); g2_dj = tap_delta(p.g_djp, z0);
    g2_toff = (g2_di * p.g_SJ + g2_dj) * p.g_C;
  }
  auto g2_advance = [&]() {
    const int step_j = BK * p.g_sj * p.g_C;
    const int wrap_j = p.g_si * p.g_SJ * p.g_C - p.g_nJ * p.g_sj * p.g_C;
    const int wrap_i = p.g_SI * p.g_SJ * p.g_C - p.g_nI * p.g_si * p.g_SJ * p.g_C;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      g2row[i] += BK; g2j[i] += BK; g2off[i] += step_j;
      while (g2j[i] >= p.g_nJ) { g2j[i] -= p.g_nJ; ++g2i[i]; g2off[i] += wrap_j; }
      while (g2i[i] >= p.g_nI) { g2i[i] -= p.g_nI; g2off[i] += wrap_i; }
    }
  };
  // G == 2 with row_len: a K-tile whose 64 positions all lie beyond their utterance's length multiplies zeros (dY is masked there):
  // staged from the zero page, neither read nor multiplied.  Evaluated once per tile in increasing order (the utterance of the
  // tile's first position is tracked incrementally); one utterance per tile only.
  int u_b = 0, u_row0 = 0, u_valid = 0;
  if constexpr (G == 2) {
    if (p.row_len) {
      u_b = (kt0 * BK) / p.rows_per_b;
      u_row0 = u_b * p.rows_per_b;
      u_valid = (int)min((long long)p.rows_per_b, p.row_len[u_b] * p.rows_inner);
    }
  }
  auto tile_live = [&](const int ts) -> bool {  // uniform
    if constexpr (G != 2) return true;
    if (!p.row_len) return true;
    const int k0 = (kt0 + ts) * BK, k1 = min(k0 + BK, p.K) - 1;
    while (k0 >= u_row0 + p.rows_per_b) {
      ++u_b; u_row0 += p.rows_per_b;
      u_valid = (int)min((long long)p.rows_per_b, p.row_len[u_b] * p.rows_inner);
    }
    return !(k1 < u_row0 + p.rows_per_b && k0 - u_row0 >= u_valid);
  };
  // half-tile j of K-tile ts (staging order j: 0 = B0, 1 = A0, 2 = B1, 3 = A1; j is a literal at every call site)
  auto stage = [&](const int j, const int ts, const bool live, const int buf = -1) __attribute__((always_inline)) {
    const int h = j >> 1;
    const bool isA = (j & 1) != 0;
    char* dst = lds8 + (buf < 0 ? (ts & 1) : buf) * BUF_B + ((isA ? 0 : AH) + h) * V8_HALF_B + wave * 1024;
    if constexpr (!TN) {
      if (G == 1 && isA) {
        const char* src = Ab + g_toff;
        const uint32_t bit = 1u << g_tap;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const char* sp = (gin[h][i] & bit) ? src + oA[h][i] : zpage;
          __builtin_amdgcn_global_load_lds((glb_void_t*)sp, (lds_void_t*)(dst + i * 8192), 16, 0, 0);
        }
      } else {
        const char* src = (isA ? Ab : Bb) + (long long)ts * (BK * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
          __builtin_amdgcn_global_load_lds((glb_void_t*)(src + (isA ? oA[h][i] : oB[h][i])), (lds_void_t*)(dst + i * 8192), 16, 0, 0);
      }
    } else {
      const bool tail = (kt0 + ts == nk_total - 1) && krem < BK;  // uniform
      if (G == 2 && !isA) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int si = g2i[i] * p.g_si + g2_di, sj = g2j[i] * p.g_sj + g2_dj;
          const bool ok = live && ((okbits >> (4 + h * 2 + i)) & 1u) && g2row[i] < p.K && (unsigned)si < (unsigned)p.g_SI &&
                          (unsigned)sj < (unsigned)p.g_SJ;
          const char* sp = ok ? (const char*)p.B + (long long)(g2off[i] + g2_toff) * 2 + oB[h][i] : zpage;
          __builtin_amdgcn_global_load_lds((glb_void_t*)sp, (lds_void_t*)(dst + i * 8192), 16, 0, 0);
        }
      } else {
        const char* src = isA ? Ab + ts * stepA : Bb + ts * stepB;
        const int sh = (isA ? 0 : 4) + h * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bool ok = live && ((okbits >> (sh + i)) & 1u) && (!tail || k_lo + 32 * i < krem);
          const char* sp = ok ? src + (isA ? oA[h][i] : oB[h][i]) : zpage;
          __builtin_amdgcn_global_load_lds((glb_void_t*)sp, (lds_void_t*)(dst + i * 8192), 16, 0, 0);
        }
      }
    }
  };

  // ---- fragment read addresses in K-tile buffer 0
  //   NT: per k-step (the chunk swizzle is an XOR); A halves / m fragments and B halves / n fragments are compile-time offsets
  //   TN: per m / n fragment (the fragment's column chunk enters the swizzle XOR); halves, k-steps and the fragment's second
  //       transpose read (k + 4) are compile-time offsets
  const int l15 = lane & 15, l4 = lane >> 4;
  const uint32_t sb8 = lds_addr(smem8);
  uint32_t fra[4], frb[2];
  if constexpr (!TN) {
    const int sw = (lane >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      fra[ks] = sb8 + (uint32_t)((wm * 64 + l15) * 128 + (((4 * ks + l4) ^ sw) << 4));
      frb[ks] = sb8 + (uint32_t)(AH * V8_HALF_B + (wn * 32 + l15) * 128 + (((4 * ks + l4) ^ sw) << 4));
    }
    fra[2] = fra[3] = 0u;
  } else {
    const int t = l15, kr = l4 * 8 + (t >> 2);
    const int f = ((t >> 2) << 2) ^ ((l4 & 1) << 1);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
      fra[mi] = sb8 + (uint32_t)(kr * 256 + (((wm * 8 + mi * 2 + ((t & 3) >> 1)) ^ f) << 4) + (t & 1) * 8);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
      frb[ni] = sb8 + (uint32_t)(2 * V8_HALF_B + kr * 256 + (((wn * 4 + ni * 2 + ((t & 3) >> 1)) ^ f) << 4) + (t & 1) * 8);
  }
  FragU fa[4][2], fb[2][2][2];  // [m fragment][k-step], [B half][n fragment][k-step]

#define V8_FRAG_A(H, MI, KS)                                                                                  \
  if constexpr (TN) {                                                                                         \
    tr_rd<(H) * V8_HALF_B + (KS) * 8192>(fa[MI][KS].h[0], ra[MI]);                                            \
    tr_rd<(H) * V8_HALF_B + (KS) * 8192 + 1024>(fa[MI][KS].h[1], ra[MI]);                                     \
  } else v8_rd128<(H) * V8_HALF_B + (MI) * 2048>(fa[MI][KS].v, ra[KS])
#define V8_FRAG_B(HB, NI, KS)                                                                                 \
  if constexpr (TN) {                                                                                         \
    tr_rd<(HB) * V8_HALF_B + (KS) * 8192>(fb[HB][NI][KS].h[0], rb[NI]);                                       \
    tr_rd<(HB) * V8_HALF_B + (KS) * 8192 + 1024>(fb[HB][NI][KS].h[1], rb[NI]);                                \
  } else v8_rd128<(HB) * V8_HALF_B + (NI) * 2048>(fb[HB][NI][KS].v, rb[KS])
#define V8_RD_A(H)                                                                                            \
  V8_FRAG_A(H, 0, 0); V8_FRAG_A(H, 0, 1); V8_FRAG_A(H, 1, 0); V8_FRAG_A(H, 1, 1);                             \
  V8_FRAG_A(H, 2, 0); V8_FRAG_A(H, 2, 1); V8_FRAG_A(H, 3, 0); V8_FRAG_A(H, 3, 1)
#define V8_RD_B(HB) V8_FRAG_B(HB, 0, 0); V8_FRAG_B(HB, 0, 1); V8_FRAG_B(HB, 1, 0); V8_FRAG_B(HB, 1, 1)
#define V8_PIN_A "+v"(fa[0][0].v), "+v"(fa[0][1].v), "+v"(fa[1][0].v), "+v"(fa[1][1].v), "+v"(fa[2][0].v), "+v"(fa[2][1].v), "+v"(fa[3][0].v), "+v"(fa[3][1].v)
#define V8_PIN_B(HB) "+v"(fb[HB][0][0].v), "+v"(fb[HB][0][1].v), "+v"(fb[HB][1][0].v), "+v"(fb[HB][1][1].v)
  // 16 MFMAs: quadrant (A half H, B half HB) x K = 64; the eight accumulators of a k-step are independent.  The operands are
  // SWAPPED (the instruction computes the transposed 16x16 block): a lane then holds FOUR CONSECUTIVE COLUMNS of one row of C
  // (row = lane & 15, columns 4 * (lane >> 4) + reg), which go into the epilogue's window as one ds_write_b128 -- four ds_write_b32
  // per fragment (the straight layout: 4 rows of one column) took 3 000 of the epilogue round's 5 900 cycles (profiles/r6_gemm_8phase.md)
#define V8_MM(H, HB)                                                                                        \
  __builtin_amdgcn_sched_barrier(0);                                                                        \
  if (live_cur) {                                                                                           \
  _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_)                                                       \
  _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                          \
  _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                          \
      acc[H][mi][HB][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[HB][ni][ks_].v, fa[mi][ks_].v, acc[H][mi][HB][ni], 0, 0, 0); \
  }                                                                                                         \
  __builtin_amdgcn_sched_barrier(0);                                                                        \
  __builtin_amdgcn_s_barrier();                                                                             \
  __builtin_amdgcn_sched_barrier(0)

  // bias gradient riding along with a weight gradient: column sums of the A tile, read back from LDS.  The tn workgroups of one
  // (tile_m, K slice) stage the same A tile: the 8 k-rows of each wave's k-group are dealt round-robin to (up to 8 of) them.
  const int cs_step = tn < 8 ? tn : 8;
  const bool do_colsum = TN && G == 0 && p.colsum_out != nullptr && tile_n < cs_step;
  float csum[4] = {0.f, 0.f, 0.f, 0.f};

  if constexpr (AH == 2) {
  // ---- prologue: all of K-tile 0 and three half-tiles of K-tile 1 (7 half-tiles = 14 DMA instructions per wave)
  V8_STAMP(0);
  bool live_cur = tile_live(0), live_n1 = tile_live(1), live_n2 = true;
  stage(0, 0, live_cur); stage(1, 0, live_cur); stage(2, 0, live_cur); stage(3, 0, live_cur);
  if constexpr (G == 1) gather_next();
  if constexpr (G == 2) g2_advance();
  stage(0, 1, live_n1); stage(1, 1, live_n1); stage(2, 1, live_n1);
  if constexpr (G == 2) g2_advance();
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  V8_STAMP(1);
  if (wm == 1) __builtin_amdgcn_s_barrier();  // the second wave row runs one barrier behind the first from here on
  __builtin_amdgcn_sched_barrier(0);

#pragma nounroll
  for (int t = 0; t < nk; ++t) {
    const uint32_t bo = (uint32_t)(t & 1) * V8_BUF_B;
    const uint32_t ra[4] = {fra[0] + bo, fra[1] + bo, fra[2] + bo, fra[3] + bo};
    const uint32_t rb[2] = {frb[0] + bo, frb[1] + bo};
    // -- phase 0: B0 (first: retired by the counted lgkmcnt, its buffer is re-staged in phase 1) + A0 -> quadrant (0, 0)
    if (live_cur) {
      V8_RD_B(0);
      V8_RD_A(0);
    }
    if (t + 1 < nk) stage(3, t + 1, live_n1);
    if constexpr (TN) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");  // 24 reads issued, the 8 of B0 are the oldest
    else asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                 // 12 reads issued, the 4 of B0 are the oldest
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" : V8_PIN_A, V8_PIN_B(0) : : "memory");
    V8_MM(0, 0);
    // -- phase 1: B1 -> quadrant (0, 1)
    if (live_cur) { V8_RD_B(1); }
    if (t + 2 < nk) {
      live_n2 = tile_live(t + 2);
      stage(0, t + 2, live_n2);
    }
    if (do_colsum && live_cur) {
      // thread -> (k-group of 8 rows = wave, A half lane >> 5, 4 consecutive columns); both A halves of tile t are still whole
      // (A0 is re-staged in phase 2: these reads retire at once, lds_rd64_sync waits for them)
      const int col = (lane & 31) * 4;
      const uint32_t a_addr = sb8 + bo + (uint32_t)((lane >> 5) * V8_HALF_B);
      for (int kr = tile_n; kr < 8; kr += cs_step) {
        const int krow = wave * 8 + kr;
        const int off = krow * 256 + ((((col >> 3) ^ ((krow & 3) << 2) ^ ((wave & 1) << 1))) << 4) + (col & 7) * 2;
        const u32x2 v = lds_rd64_sync(a_addr + (uint32_t)off);
        csum[0] += __uint_as_float(v[0] << 16); csum[1] += __uint_as_float(v[0] & 0xffff0000u);
        csum[2] += __uint_as_float(v[1] << 16); csum[3] += __uint_as_float(v[1] & 0xffff0000u);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" : V8_PIN_B(1) : : "memory");
    V8_MM(0, 1);
    // -- phase 2: A1 -> quadrant (1, 1)
    if (live_cur) { V8_RD_A(1); }
    if (t + 2 < nk) {
      if constexpr (G == 1) gather_next();
      stage(1, t + 2, live_n2);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" : V8_PIN_A : : "memory");
    V8_MM(1, 1);
    // -- phase 3: quadrant (1, 0) from registers; the K-tile's one DMA wait: tile t+1 complete, tile t+2's three half-tiles in flight
    if (t + 2 < nk) {
      stage(2, t + 2, live_n2);
      if constexpr (G == 2) g2_advance();
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    V8_MM(1, 0);
    live_cur = live_n1; live_n1 = live_n2;
  }
  } else {
  // ---- one A half: prologue = K-tiles 0 and 1 (6 half-tiles = 12 DMA instructions per wave)
  V8_STAMP(0);
  bool live_cur = true;
  stage(0, 0, true, 0); stage(1, 0, true, 0); stage(2, 0, true, 0);
  stage(0, 1, true, 1); stage(1, 1, true, 1); stage(2, 1, true, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  V8_STAMP(1);
  if (wm == 1) __builtin_amdgcn_s_barrier();  // the second wave row runs one barrier behind the first from here on
  __builtin_amdgcn_sched_barrier(0);
  int bcur = 0, bnew = 2;  // buffers of tile t and of tile t+2
#pragma nounroll
  for (int t = 0; t < nk; ++t) {
    const uint32_t bo = (uint32_t)bcur * BUF_B;
    const uint32_t ra[4] = {fra[0] + bo, fra[1] + bo, 0u, 0u};
    const uint32_t rb[2] = {frb[0] + bo, frb[1] + bo};
    // -- phase 0: B0 + A0 -> quadrant (0, 0); B0 of tile t+2 goes where tile t-1's was (last read two phases ago)
    V8_RD_B(0);
    V8_RD_A(0);
    if (t + 2 < nk) stage(0, t + 2, true, bnew);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" : V8_PIN_A, V8_PIN_B(0) : : "memory");
    V8_MM(0, 0);
    // -- phase 1: B1 -> quadrant (0, 1); B1 and A0 of tile t+2; the K-tile's one DMA wait: tile t+1 complete, tile t+2 in flight
    V8_RD_B(1);
    if (t + 2 < nk) {
      stage(2, t + 2, true, bnew);
      stage(1, t + 2, true, bnew);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" : V8_PIN_B(1) : : "memory");
    V8_MM(0, 1);
    bcur = bcur == 2 ? 0 : bcur + 1;
    bnew = bnew == 2 ? 0 : bnew + 1;
  }
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();  // re-join the wave rows
#undef V8_FRAG_A
#undef V8_FRAG_B
#undef V8_RD_A
#undef V8_RD_B
#undef V8_PIN_A
#undef V8_PIN_B
#undef V8_MM
  __syncthreads();
  V8_STAMP(2);

  float* sC = reinterpret_cast<float*>(smem8);
  if (do_colsum) {  // combine the 8 k-groups (waves) through LDS: 8 x 256 floats; column = (lane >> 5) * 128 + (lane & 31) * 4 + e
#pragma unroll
    for (int e = 0; e < 4; ++e) sC[wave * BM2 + lane * 4 + e] = csum[e];
    __syncthreads();
    if (threadIdx.x < BM2) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += sC[w * BM2 + threadIdx.x];
      const int m = m0 + threadIdx.x;
      if (m < p.M) atomicAdd(p.colsum_out + z0 * p.colsum_stride + m, v);
    }
    __syncthreads();
  }

  // ---- epilogue: four rounds (A half h, m-fragment pair pr) of 64 rows x 256 columns through an f32 window in LDS: window row
  // wm*32 + mi2*16 + (lane & 15) = tile row h*128 + (rl >> 5)*64 + pr*32 + (rl & 31); a lane's four registers are columns
  // 4*(lane >> 4) + 0..3 of the fragment (swapped operands, see V8_MM): one 16-byte LDS store per fragment.
  const bool fast = (p.vec_ok & 1) && !(p.N & 7) && n0 + BN4 <= p.N && !p.atomic;
#define V8_PICK(H, M0)                                                                          \
  _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)   \
  _Pragma("unroll") for (int c = 0; c < 2; ++c) tq[a][b][c] = acc[H][M0 + a][b][c];
#define V8_PICK_ROUND()                                                    \
  switch (r) {                                                             \
    case 0: V8_PICK(0, 0) break;                                           \
    case 1: V8_PICK(0, 2) break;                                           \
    default:                                                               \
      if constexpr (AH == 2) {                                             \
        if (r == 2) { V8_PICK(1, 0) } else { V8_PICK(1, 2) }               \
      }                                                                    \
      break;                                                               \
  }
  if (fast) {
    // full-width vector epilogue: TWO swizzled [64][256] windows (a round's writes go to the window the round before the previous one
    // read: one barrier per round), v8_round_fast on top
    float b8[8];
    if (p.bias) ld8x(p.bias, n0 + (threadIdx.x & 31) * 8, MI_DT_F32, b8);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) b8[j] = 0.f;
    }
#pragma nounroll
    for (int r = 0; r < 2 * AH; ++r) {
      f32x4 tq[2][2][2];  // [mi2][hb][ni]; selected by a uniform switch: the accumulators are never indexed dynamically
      V8_PICK_ROUND()
#if defined(V8_EPI_ABL) && V8_EPI_ABL == 1
      asm volatile("" :: "v"(tq[0][0][0]), "v"(tq[0][0][1]), "v"(tq[0][1][0]), "v"(tq[0][1][1]), "v"(tq[1][0][0]), "v"(tq[1][0][1]), "v"(tq[1][1][0]), "v"(tq[1][1][1]));
      continue;
#endif
      float* win = sC + (r & 1) * (64 * 256);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int c = 0; c < 2; ++c)
            *reinterpret_cast<f32x4*>(win + (wm * 32 + a * 16 + l15) * 256 + (((b * 32 + wn * 8 + c * 4 + l4) ^ (l15 & 7)) << 2)) = tq[a][b][c];
#if defined(V8_EPI_ABL) && V8_EPI_ABL == 2
      continue;
#endif
      __syncthreads();
      const int mb = m0 + (r >> 1) * 128 + (r & 1) * 32;
      switch (p.epi) {
        case EPI_STORE: v8_round_fast<EPI_STORE>(p, win, b8, z, coff, mb, 64, n0); break;
        case EPI_SWISH_DROP: v8_round_fast<EPI_SWISH_DROP>(p, win, b8, z, coff, mb, 64, n0); break;
        case EPI_RESID: v8_round_fast<EPI_RESID>(p, win, b8, z, coff, mb, 64, n0); break;
        case EPI_DSWISH: v8_round_fast<EPI_DSWISH>(p, win, b8, z, coff, mb, 64, n0); break;
        case EPI_RELU_MASK: v8_round_fast<EPI_RELU_MASK>(p, win, b8, z, coff, mb, 64, n0); break;
        default: v8_round_fast<EPI_MUL_POS>(p, win, b8, z, coff, mb, 64, n0); break;
      }
    }
  } else {
    // partial column tiles, unaligned or column-strided C, split-K atomics: the padded [64][260] window and round-out of the third structure
    constexpr int LDS_C = BN4 + 4;
#pragma nounroll
    for (int r = 0; r < 2 * AH; ++r) {
      f32x4 tq[2][2][2];
      V8_PICK_ROUND()
      if (r) __syncthreads();
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int c = 0; c < 2; ++c)
            *reinterpret_cast<f32x4*>(sC + (wm * 32 + a * 16 + l15) * LDS_C + b * 128 + wn * 32 + c * 16 + l4 * 4) = tq[a][b][c];
      __syncthreads();
      v4_round_out(p, sC, z, coff, m0, n0, (r >> 1) * 128 + (r & 1) * 32, false, 64);
    }
  }
#undef V8_PICK_ROUND
#undef V8_PICK
#ifdef V8_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the stores have left the CU)
  __syncthreads();
  V8_STAMP(3);
#endif
}

template <int G, bool TN, int AH = 2>
__global__ __launch_bounds__(512) void gemm_bf16_v8_kernel(GemmP p) {
  drop_resolve(p.drop);
  extern __shared__ __attribute__((aligned(16))) bf16_t smem8[];  // 2 K-tiles x 4 half-tile images x 16 KiB (AH = 1: 3 x 3 x 16 KiB)
  const int tn = (p.N + BN4 - 1) / BN4, tm = (p.M + 128 * AH - 1) / (128 * AH);
  const int ntiles = tm * tn;
  int logical, z, ks;
  if constexpr (G == 2) {
    // conv weight gradient (taps x tiles x split-K): the (K slice, tap, tile) list is laid out slice-major and every XCD takes a
    // contiguous eighth of it, so that co-resident workgroups work on the same slice of dY / the activation grid (third structure)
    const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
    const int Wt = gx * gy * gz;
    const int L = (int)blockIdx.x + gx * ((int)blockIdx.y + gy * (int)blockIdx.z);  // dispatch order: XCD = L % 8
    const int q8w = Wt >> 3, r8w = Wt & 7, xw = L & 7;
    const int w = (xw < r8w ? xw * (q8w + 1) : r8w * (q8w + 1) + (xw - r8w) * q8w) + (L >> 3);
    const int ncombo = gx * gz;
    ks = w / ncombo;
    const int combo = w - ks * ncombo;
    z = combo / gx;
    logical = combo - z * gx;
  } else {
    const int bid = blockIdx.x;
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    z = blockIdx.z;
    ks = blockIdx.y;
  }
  // tile order inside the list: column blocks of (up to) 8 tiles, row-major inside a block -- the 32 workgroups an XCD runs
  // together then cover about 4 x 8 tiles (12 operand panels through its L2) instead of 1 x 32 (33 panels) when N is wide
  // (the Conformer shapes have tn <= 8 and keep their order)
  const int per_cb = tm * 8;
  const int cb = logical / per_cb, rem = logical - cb * per_cb;
  const int w8 = min(8, tn - cb * 8);
  const int tile_m = rem / w8, tile_n = cb * 8 + (rem - tile_m * w8);
  // Phase offset between two halves of the chip (short-K problems with more than one round of tiles): all 256 workgroups of a round
  // otherwise finish their K loops together and write their tiles together -- the fabric takes a 33-65 MB burst at ~4.5 TB/s while
  // every matrix pipe idles, then idles itself through the next K loops (timeline: profiles/r6_gemm_8phase.md).  Every other group
  // of 8 first-round workgroups starts `v8_delay` ticks late, so that one half multiplies while the other half writes.
  if (p.v8_delay > 0 && blockIdx.x < 256 && (blockIdx.x & 8)) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)p.v8_delay) __builtin_amdgcn_s_sleep(16);
  }
  gemm_v8_body<G, TN, AH>(p, smem8, tile_m, tile_n, ks, z);
}

// ---- grouped weight gradients on the eighth structure: the problems' 256x256 tiles x K slices in one launch, (problem, K slice)
// groups laid out one after the other and dealt to the XCDs in contiguous eighths (as gemm_bf16_grouped_tn_kernel above)
__global__ __launch_bounds__(512) void gemm_bf16_grouped_tn8_kernel(GroupP g) {
  extern __shared__ __attribute__((aligned(16))) bf16_t smem8[];
  int pi = 0;
  const int tiles_total = g.tile_begin[g.n];
  const int W = tiles_total * g.splitk;
  const int L = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
  const int q8w = W >> 3, r8w = W & 7, xw = L & 7;
  const int w = (xw < r8w ? xw * (q8w + 1) : r8w * (q8w + 1) + (xw - r8w) * q8w) + (L >> 3);
  while (pi + 1 < g.n && w >= g.splitk * g.tile_begin[pi + 1]) ++pi;  // uniform
  const int ntile_p = g.tile_begin[pi + 1] - g.tile_begin[pi];
  const int rel = w - g.splitk * g.tile_begin[pi];
  const int exp_ks = rel / ntile_p, exp_tile = rel - exp_ks * ntile_p;
  GemmP p;
  p.A = g.A[pi]; p.B = g.B[pi]; p.C = g.C[pi];
  p.M = g.M[pi]; p.N = g.N[pi]; p.K = g.K;
  p.lda = g.lda[pi]; p.ldb = g.ldb[pi]; p.ldc = g.ldc[pi]; p.csc = 1;
  p.transA = 1; p.transB = 1; p.batch = 1; p.nb0 = 1;
  p.sA0 = p.sA1 = p.sB0 = p.sB1 = p.sC0 = p.sC1 = 0;
  p.bias = nullptr; p.alpha = 1.f; p.epi = EPI_STORE; p.c_dt = MI_DT_F32; p.atomic = 1; p.swish_g = 0;
  p.aux_in = nullptr; p.auxin_dt = 0; p.aux_out = nullptr; p.auxout_dt = 0; p.ldaux = 0;
  p.drop.key = 0u; p.drop.threshold = 0u; p.drop.scale = 1.f; p.drop.step = nullptr;
  p.row_len = nullptr; p.rows_per_b = 1; p.rows_inner = 1;
  p.splitk = g.splitk; p.ktiles_per_split = g.ktiles_per_split;
  p.colsum_stride = 0; p.colsum_out = g.colsum[pi];
  p.vec_ok = 0; p.g_on = 0; p.r_on = 0; p.v8_delay = 0;
  const int tn = (p.N + BN4 - 1) / BN4;
  gemm_v8_body<0, true>(p, smem8, exp_tile / tn, exp_tile % tn, exp_ks, 0);
}

// =================================================================================================
// Fifth structure: the FFN / projection GEMMs of the Conformer block (NT, K = 512 ... 2048, full-width epilogues).
// The 256x128 / 256x256 structures above run K loop and epilogue strictly one after the other -- one workgroup owns the
// CU and all workgroups move in step, so the HBM idles during the K loops and the matrix pipe during the epilogues; at
// K = 512 the epilogue (LDS round trip + Swish / dropout VALU work + 130 MB of stores) costs twice the K loop.  Here:
//   * PERSISTENT workgroups (one per CU, 512 threads), each walking its share of the 256x128 output tiles;
//   * TWO accumulator sets per wave (64 + 64 registers): while tile t+1 runs its K loop, tile t's epilogue proceeds in
//     eight 8-row ROUNDS, one per K-tile iteration;
//   * the rounds go through a WAVE-PRIVATE LDS window (8 rows x 64 columns f32, XOR-swizzled so that ds_write_b32 and
//     ds_read_b64 are conflict-free): accumulator layout (lane = column) -> row layout (lane = 8 consecutive columns), so
//     global accesses are 16/32 B per lane and 128/256 B per row, no workgroup barrier involved;
//   * fragment reads run one k-step ahead of the MFMAs (inline asm, counted lgkmcnt), aux_in (residual / pre-activation) of a
//     round is fetched one iteration ahead by inline-asm loads covered by the iteration's counted vmcnt -- an ordinary load
//     or a compiler-visible LDS read next to LDS-DMA in flight makes hipcc wait vmcnt(0) and drain the operand pipeline;
//   * the LDS-DMA ring (three 48-KiB stages, two K-tiles ahead, counted vmcnt) runs ACROSS tile boundaries: the first two
//     K-tiles of the next tile are in flight during the last two iterations of the current one.
// LDS: 3 x 48 KiB + 8 x 2 KiB windows = exactly the 160 KiB of the CU.  Same products in the same k order as the tiled
// structures; the bias enters the sum first (accumulators start from it) instead of last.
// What it buys, measured same-box against the 256x256 / 256x128 structures (profiles/r2_gemm_structures.md): the window
// transposition is free (hidden), the stores cost 4-9 us and the Swish / dropout arithmetic 19 us of a 63-us FFN1 launch
// whether or not a SIMD's second wave is multiplying meanwhile (running the two wave halves in opposite order was 8 %
// SLOWER); net -6 % on the Swish-gradient epilogue, -3 % on N = 1536 stores, +-0 on Swish forward, slower for K >= 1024
// (its K loop runs at the 256x128 rate).  The dispatcher uses it where it wins.
// =================================================================================================
#define V5_WIN (8 * 64)                                    // floats per wave window
#define V5_LDS_BYTES (3 * NT2_STAGE * 2 + 8 * V5_WIN * 4)  // 163 840

// LDS byte addresses of a lane's window accesses (loop-invariant, 8 registers).  The window accesses are INLINE ASM: the
// compiler orders every LDS read it can see behind ALL LDS-DMA in flight (it cannot tell the window from the operand
// stages) and would put `s_waitcnt vmcnt(0)` in front of the window reads -- draining the operand pipeline every iteration.
// One wave's DS operations execute in order, so its reads see its own writes without a wait in between.
struct V5Win { uint32_t wa[4], ra[4]; int oc, oa, od; };  // + per-lane element offsets (row lane>>3, column chunk) into C / aux / index space
__device__ __forceinline__ void v5_win_setup(V5Win& w, const GemmP& p, const float* win, const int lane) {
  w.oc = (lane >> 3) * (int)p.ldc + (lane & 7) * 8;
  w.oa = (lane >> 3) * (int)p.ldaux + (lane & 7) * 8;
  w.od = (lane >> 3) * p.N + (lane & 7) * 8;
  const uint32_t base = (uint32_t)(uintptr_t)(lds_void_t*)win;
  const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int e = 0; e < 4; ++e) w.wa[e] = base + 4u * (uint32_t)((4 * lh + e) * 64 + (lr ^ (2 * e)));  // column 32 + lr: +128 B
  const int wr = lane >> 3, c8 = (lane & 7) * 8, sw = 2 * (wr & 3);
#pragma unroll
  for (int k = 0; k < 4; ++k) w.ra[k] = base + 4u * (uint32_t)(wr * 64 + ((c8 + 2 * k) ^ sw));
}

// aux_in (residual stream, f32 / Swish pre-activation, bf16) of one round, fetched ONE ITERATION AHEAD by inline-asm loads:
// an ordinary load would make the compiler wait `vmcnt(0)` at its first use inside the round -- draining the operand DMA.
// The loads are issued ahead of the iteration's DMA, so the counted wait that ends the iteration covers them.
template <int EPI>
__device__ __forceinline__ void v5_aux_issue(const GemmP& p, u32x4 (&aux)[2], const int r, const int m_blk, const int n_blk,
                                             const int lane) {
  if (EPI != EPI_RESID && EPI != EPI_DSWISH) return;
  // ONE unconditional issue site per variable, operands tied in/out ("+v"): the compiler believes an asm output is valid at
  // the end of the statement, so any copy it makes of it (merging two issue sites, an exec-masked branch ...) happens BEFORE
  // the data lands and leaves a register the load later overwrites (seen: memory faults).  Rows past M read row M-1.
  int m = m_blk + (r >> 2) * 32 + 8 * (r & 3) + (lane >> 3);
  m = m < p.M ? m : p.M - 1;
  const long long ai = (long long)m * p.ldaux + n_blk + (lane & 7) * 8;
  if (EPI == EPI_RESID) {
    const float* src = (const float*)p.aux_in + ai;
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"
                 : "+v"(aux[0]), "+v"(aux[1]) : "v"(src) : "memory");
  } else {
    const bf16_t* src = (const bf16_t*)p.aux_in + ai;
    asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(aux[0]) : "v"(src) : "memory");
  }
}

// one 8-row epilogue round of a wave's 64x64 block of the PREVIOUS tile.  r = 0..7: accumulators [r>>2][*], registers
// 4*(r&3) .. +3 of each  <->  block rows (r>>2)*32 + 8*(r&3) + 0..7
template <int EPI>
__device__ __forceinline__ void v5_round(const GemmP& p, const V5Win& w, const f32x16 (&acc)[2][2], const int r, const int m_blk,
                                         const int n_blk, const u32x4 (&aux)[2], const int lane) {
  f32x4 t0, t1;  // columns lr (j = 0) and 32 + lr (j = 1), rows 4*lh + 0..3 of the round
#define V5_PICK(i, q)                                                                                       \
  t0 = f32x4{acc[i][0][4 * q], acc[i][0][4 * q + 1], acc[i][0][4 * q + 2], acc[i][0][4 * q + 3]};         \
  t1 = f32x4{acc[i][1][4 * q], acc[i][1][4 * q + 1], acc[i][1][4 * q + 2], acc[i][1][4 * q + 3]};
  switch (r) {  // uniform: the accumulators are never indexed dynamically
    case 0: V5_PICK(0, 0) break;
    case 1: V5_PICK(0, 1) break;
    case 2: V5_PICK(0, 2) break;
    case 3: V5_PICK(0, 3) break;
    case 4: V5_PICK(1, 0) break;
    case 5: V5_PICK(1, 1) break;
    case 6: V5_PICK(1, 2) break;
    default: V5_PICK(1, 3) break;
  }
#undef V5_PICK
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(w.wa[e]), "v"(t0[e]) : "memory");
    asm volatile("ds_write_b32 %0, %1 offset:128" ::"v"(w.wa[e]), "v"(t1[e]) : "memory");
  }
  // row layout: lane -> window row lane>>3, columns 8*(lane&7) .. +7 (four conflict-free ds_read_b64)
  const int wr = lane >> 3, c8 = (lane & 7) * 8;
  float v[8];
  {
    mi_f32x2 x0, x1, x2, x3;
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)
                 : "v"(w.ra[0]), "v"(w.ra[1]), "v"(w.ra[2]), "v"(w.ra[3])
                 : "memory");
    v[0] = x0[0]; v[1] = x0[1]; v[2] = x1[0]; v[3] = x1[1]; v[4] = x2[0]; v[5] = x2[1]; v[6] = x3[0]; v[7] = x3[1];
  }
  // addresses: (uniform first row of the round) x pitch on the scalar unit + a loop-invariant per-lane offset -- no
  // per-lane 64-bit multiplies (quarter rate); M x pitch < 2^31 by contract
  const int mu = __builtin_amdgcn_readfirstlane(m_blk + (r >> 2) * 32 + 8 * (r & 3));
  if (mu + wr >= p.M) return;
  const long long ci = (long long)(mu * (int)p.ldc + n_blk + w.oc), ai = (long long)(mu * (int)p.ldaux + n_blk + w.oa);
  float dm[8];  // (the bias is already in the accumulators: they start from it, see the tile switch)
  drop_mask8(p.drop, (uint32_t)(mu * p.N + n_blk + w.od), dm);
  if (EPI == EPI_STORE) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= p.alpha * dm[j];
  } else if (EPI == EPI_SWISH_DROP) {
    if (p.swish_g) {
      float g[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) swish_pair(v[j], dm[j], v[j], g[j]);
      st8x(p.aux_out, ai, p.auxout_dt, g);
    } else {
      st8x(p.aux_out, ai, p.auxout_dt, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = swishf_(v[j]) * dm[j];
    }
  } else if (EPI == EPI_RESID) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = __uint_as_float(aux[j >> 2][j & 3]) + p.alpha * v[j] * dm[j];
  } else if (p.swish_g) {  // EPI_DSWISH, aux_in = g (bf16): one multiply per element
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[2 * j] *= __uint_as_float(aux[0][j] << 16);
      v[2 * j + 1] *= __uint_as_float(aux[0][j] & 0xffff0000u);
    }
  } else {  // EPI_DSWISH (aux_in bf16)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[2 * j] = v[2 * j] * dm[2 * j] * swish_grad(__uint_as_float(aux[0][j] << 16));
      v[2 * j + 1] = v[2 * j + 1] * dm[2 * j + 1] * swish_grad(__uint_as_float(aux[0][j] & 0xffff0000u));
    }
  }
  st8x(p.C, ci, p.c_dt, v);
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_v5_kernel(GemmP p, const int ntiles, const int tn) {
  drop_resolve(p.drop);
  extern __shared__ __attribute__((aligned(16))) bf16_t smem5[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // store instructions per lane and round (16 bytes each): C, plus the pre-activation of the Swish epilogue
  const int st_per_round = (p.c_dt == MI_DT_F32 ? 2 : 1) + (EPI == EPI_SWISH_DROP ? (p.auxout_dt == MI_DT_F32 ? 2 : 1) : 0);
  V5Win W5;
  v5_win_setup(W5, p, reinterpret_cast<const float*>(smem5 + 3 * NT2_STAGE) + wave * V5_WIN, lane);
  // LDS byte addresses of the A fragment reads in stage 0 per k-step (XOR swizzle: non-affine in kk); B = A + pipe5_b
  uint32_t pipe5[4];
  const uint32_t pipe5_b = (uint32_t)(BM2 * BK * 2 + ((wn * 64) - (wm * 64)) * 128);
  {
    const uint32_t base = (uint32_t)(uintptr_t)(lds_void_t*)smem5;
    const int lr = lane & 31, lh = lane >> 5, x = (lr >> 1) & 7;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) pipe5[kk] = base + (uint32_t)((wm * 64 + lr) * 128) + (uint32_t)(((kk * 2 + lh) ^ x) << 4);
  }

  // tiles of this workgroup: XCD x = blockIdx & 7 owns a contiguous chunk of the (row-major) tile list, its gridDim/8
  // workgroups take the chunk's tiles round-robin -- the tiles in flight on one XCD are consecutive (same A panels, all of B)
  const int nslot = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int chunkn = q8 + (xcd < r8 ? 1 : 0);
  const int my_tiles = slot < chunkn ? (chunkn - slot + nslot - 1) / nslot : 0;
  if (my_tiles == 0) return;
  const int nk = (p.K + BK - 1) / BK;  // >= 8 by contract (one epilogue round per K-tile iteration)
  const int total = my_tiles * nk;
  const bool ktail = (p.K & (BK - 1)) != 0;
  const bf16_t* A = (const bf16_t*)p.A;
  const bf16_t* B = (const bf16_t*)p.B;

  // ---- LDS-DMA cursor: runs two K-tiles ahead of the compute cursor, across tile boundaries.  Per thread only a 32-bit
  // element offset per 16-byte chunk is kept (6 registers for both operands; M * K and N * K < 2^31 by contract)
  int oA[4], oB[2];
  int d_tile = 0, d_kt = 0, d_stage = 0;
  auto dma_next = [&]() {
    if (d_kt == 0) {
      const int L = chunk0 + slot + d_tile * nslot;
      const int tile_m = L / tn, tile_n = L - tile_m * tn;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = threadIdx.x + i * 512;
        const int r = q >> 3, gck = (q & 7) ^ ((r >> 1) & 7);
        int gr = tile_m * BM2 + r;
        gr = gr < p.M ? gr : p.M - 1;  // rows past the matrix are clamped (never stored)
        oA[i] = gr * (int)p.lda + gck * 8;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int q = threadIdx.x + i * 512;
        const int r = q >> 3, gck = (q & 7) ^ ((r >> 1) & 7);
        oB[i] = (tile_n * BN + r) * (int)p.ldb + gck * 8;  // N % 128 == 0: every row exists
      }
    }
    bf16_t* st = smem5 + d_stage * NT2_STAGE;
    const int k0 = d_kt * BK;
    const bool tail = ktail && (k0 + BK > p.K);  // uniform: only the last K-tile of a ragged K tests its chunks
    const int kin = (((threadIdx.x & 7) ^ ((threadIdx.x >> 4) & 7)) << 3);  // (q & 7) ^ ((r >> 1) & 7), same for every i
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16_t* src = A + (long long)(oA[i] + k0);
      if (tail) src = (k0 + kin < p.K) ? src : reinterpret_cast<const bf16_t*>(g_zero16);
      __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(st + (wave * 64 + i * 512) * 8), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bf16_t* src = B + (long long)(oB[i] + k0);
      if (tail) src = (k0 + kin < p.K) ? src : reinterpret_cast<const bf16_t*>(g_zero16);
      __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(st + BM2 * BK + (wave * 64 + i * 512) * 8), 16, 0, 0);
    }
    d_stage = d_stage == 2 ? 0 : d_stage + 1;
    if (++d_kt == nk) { d_kt = 0; ++d_tile; }
  };

  // two accumulator sets in FIXED registers: the tile loop is unrolled by two with the roles swapped (a copy "previous =
  // current" at the tile switch turns into register shuffling on every iteration: measured 18 VALU instructions per MFMA)
  f32x16 accA[2][2], accB[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accA[i][j][r] = 0.f; accB[i][j][r] = 0.f; }

  dma_next();
  if (total > 1) dma_next();
  if (total > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int g = 0, c_tile = 0, c_stage = 0;
  int m_cur, n_cur, m_prev = 0, n_prev = 0;
  bool have_prev = false;
  {
    const int L = chunk0 + slot;
    const int tile_m = L / tn, tile_n = L - tile_m * tn;
    m_cur = tile_m * BM2 + wm * 64; n_cur = tile_n * BN + wn * 64;
  }
  // The accumulators of a tile START from its bias (two values per lane in the accumulator layout: columns lr and 32 + lr
  // of the wave's block) instead of zero: no bias registers in the epilogue.  The next tile's pair is loaded one iteration
  // before the tile switch, ahead of that iteration's DMA.
  float bz0 = 0.f, bz1 = 0.f;
  if (p.bias) { bz0 = p.bias[n_cur + (lane & 31)]; bz1 = p.bias[n_cur + 32 + (lane & 31)]; }
  u32x4 auxc[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}}, auxn[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};

  // one tile: K loop into `acc`, the previous tile's epilogue rounds out of `accP`
  auto tile_pass = [&](auto& acc, auto& accP) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][0][r] = bz0; acc[i][1][r] = bz1; }
    for (int it = 0; it < nk; ++it, ++g) {
      const bool more2 = g + 2 < total;
      // aux_in of the round that runs in the NEXT iteration: round it+1 of the previous tile, or round 0 of the tile that
      // is finishing in this iteration
      {
        int ar = -1, am = 0, an = 0;
        if (it + 1 == nk) { if (g + 1 < total) { ar = 0; am = m_cur; an = n_cur; } }
        else if (have_prev && it + 1 < 8) { ar = it + 1; am = m_prev; an = n_prev; }
        if (ar >= 0) v5_aux_issue<EPI>(p, auxn, ar, am, an, lane);
      }
      if (p.bias && it == nk - 1 && c_tile + 1 < my_tiles) {
        const int L = chunk0 + slot + (c_tile + 1) * nslot;
        const int nn = (L % tn) * BN + wn * 64 + (lane & 31);
        bz0 = p.bias[nn]; bz1 = p.bias[nn + 32];
      }
      __builtin_amdgcn_sched_barrier(0);  // (the DMA count below assumes nothing slips behind the DMA issue)
      if (more2) dma_next();  // stage of iteration g-1: every wave passed the barrier after reading it
      __builtin_amdgcn_sched_barrier(0);
      const bool round = have_prev && it < 8;
      {
        // fragment reads one k-step ahead of the MFMAs, inline asm with counted waits (see the 256x256 structure)
        const uint32_t sb = (uint32_t)(c_stage * (NT2_STAGE * 2));
        bf16x8 fa[2][2], fb[2][2];
#define V5_RD(S, KK)                                                                                              \
  {                                                                                                               \
    const uint32_t aa = pipe5[KK] + sb, ab = aa + pipe5_b;                                                        \
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\tds_read_b128 %2, %5\n\t"              \
                 "ds_read_b128 %3, %5 offset:4096"                                                                \
                 : "=&v"(fa[S][0]), "=&v"(fa[S][1]), "=&v"(fb[S][0]), "=&v"(fb[S][1])                             \
                 : "v"(aa), "v"(ab)                                                                               \
                 : "memory");                                                                                     \
  }
#define V5_WAIT(N, S)                                                                                             \
  asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(fa[S][0]), "+v"(fa[S][1]), "+v"(fb[S][0]), "+v"(fb[S][1]) : : "memory"); \
  __builtin_amdgcn_sched_barrier(0)
#define V5_MM(S)                                                                                                  \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                     \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[S][i], fb[S][j], acc[i][j], 0, 0, 0);                \
  __builtin_amdgcn_sched_barrier(0)
        V5_RD(0, 0);
        V5_RD(1, 1); V5_WAIT(4, 0); V5_MM(0);
        V5_RD(0, 2); V5_WAIT(4, 1); V5_MM(1);
        V5_RD(1, 3); V5_WAIT(4, 0); V5_MM(0);
        V5_WAIT(0, 1); V5_MM(1);
#undef V5_RD
#undef V5_WAIT
#undef V5_MM
      }
      __builtin_amdgcn_sched_barrier(0);
      if (round) v5_round<EPI>(p, W5, accP, it, m_prev, n_prev, auxc, lane);
      __builtin_amdgcn_sched_barrier(0);
      // tile g+1 must have landed (this wave's share) before the barrier publishes it.  vmcnt counts in order and counts
      // stores: younger than that tile's DMA are the 6 DMA instructions of tile g+2 (if issued) and the store
      // instructions this wave issued in this iteration's round -- `nst` of them, none when the round's 8 rows lie past M
      // (the branch around an all-inactive store is taken).  Waiting for FEWER outstanding operations than that is always
      // safe, for more never.
      {
        const int nst = (round && m_prev + (it >> 2) * 32 + 8 * (it & 3) < p.M) ? st_per_round : 0;
        if (more2) {
          switch (nst) {
            case 0: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
          }
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (EPI == EPI_RESID || EPI == EPI_DSWISH) {
        // the prefetched aux_in has landed (it is older than the DMA just waited for); the asm touch pins the register
        // copy behind the wait (a register-only move may otherwise be scheduled above an asm statement)
        asm volatile("" : "+v"(auxn[0]), "+v"(auxn[1]));
        auxc[0] = auxn[0]; auxc[1] = auxn[1];
      }
      __builtin_amdgcn_s_barrier();
      c_stage = c_stage == 2 ? 0 : c_stage + 1;
    }
    // tile finished: its accumulators are the "previous" set of the next pass
    m_prev = m_cur; n_prev = n_cur; have_prev = true;
    ++c_tile;
    if (c_tile < my_tiles) {
      const int L = chunk0 + slot + c_tile * nslot;
      const int tile_m = L / tn, tile_n = L - tile_m * tn;
      m_cur = tile_m * BM2 + wm * 64; n_cur = tile_n * BN + wn * 64;
    }
  };
  // ---- the last tile's epilogue has nothing to hide behind
  auto tail = [&](auto& accP) __attribute__((always_inline)) {
#pragma unroll 1
    for (int r = 0; r < 8; ++r) {
      if (EPI == EPI_RESID || EPI == EPI_DSWISH) {
        v5_aux_issue<EPI>(p, auxc, r, m_prev, n_prev, lane);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(auxc[0]), "+v"(auxc[1]) : : "memory");
      }
      v5_round<EPI>(p, W5, accP, r, m_prev, n_prev, auxc, lane);
    }
  };
  while (true) {
    tile_pass(accA, accB);
    if (c_tile == my_tiles) { tail(accA); break; }
    tile_pass(accB, accA);
    if (c_tile == my_tiles) { tail(accB); break; }
  }
}

// =================================================================================================
// exact fp32 VALU kernel, arbitrary strides: 64x64x16 tile, 256 threads, 4x4 per thread
// =================================================================================================
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmP p) {
  drop_resolve(p.drop);
  __shared__ float As[16][64 + 4];
  __shared__ float Bs[16][64 + 4];
  const int tn = (p.N + 63) / 64;
  const int tile_m = blockIdx.x / tn, tile_n = blockIdx.x - tile_m * tn;
  const int m0 = tile_m * 64, n0 = tile_n * 64;
  const int z = blockIdx.z, z0 = z % p.nb0, z1 = z / p.nb0;
  const float* A = (const float*)p.A + z0 * p.sA0 + z1 * p.sA1;
  const float* B = (const float*)p.B + z0 * p.sB0 + z1 * p.sB1;
  const long long coff = z0 * p.sC0 + z1 * p.sC1;
  // element (row, k) strides
  const long long sar = p.transA ? 1 : p.lda, sak = p.transA ? p.lda : 1;
  const long long sbr = p.transB ? 1 : p.ldb, sbk = p.transB ? p.ldb : 1;
  int k_begin = 0, k_end = p.K;
  if (p.splitk > 1) {
    k_begin = blockIdx.y * p.ktiles_per_split * BK;
    k_end = min(p.K, k_begin + p.ktiles_per_split * BK);
    if (k_begin >= k_end) return;
  }
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  float acc[4][4] = {};
  for (int k0 = k_begin; k0 < k_end; k0 += 16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = threadIdx.x + i * 256;
      int r, k;
      if (p.transA) { r = e & 63; k = e >> 6; } else { k = e & 15; r = e >> 4; }
      float v = 0.f;
      if (m0 + r < p.M && k0 + k < k_end) v = A[(long long)(m0 + r) * sar + (long long)(k0 + k) * sak];
      As[k][r] = v;
      if (p.transB) { r = e & 63; k = e >> 6; } else { k = e & 15; r = e >> 4; }
      v = 0.f;
      if (n0 + r < p.N && k0 + k < k_end) v = B[(long long)(n0 + r) * sbr + (long long)(k0 + k) * sbk];
      Bs[k][r] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[k][ty * 4 + i]; b[i] = Bs[k][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
      if (m < p.M && n < p.N) epilogue(p, z, coff, m, n, acc[i][j]);
    }
}

// =================================================================================================
// exact fp32 on the matrix cores: v_mfma_f32_32x32x2_f32 (fp32 operands, fp32 accumulation -- no reduced-precision path, unlike
// xf32), same 64x64x16 tile / staging / scalar epilogue / arbitrary strides as the VALU kernel above; 4 waves, one 32x32
// quadrant each, 8 MFMAs per K-tile.  Operand layout of the instruction: lane l supplies A[i = l & 31][k = l >> 5] and
// B[k = l >> 5][j = l & 31]; D[i][j] comes back with j = l & 31, i = (r & 3) + 8 (r >> 2) + 4 (l >> 5) for register r.
// The fp32 path is the parity configuration (BASELINE configs[0], every fp32 test): 256 FLOP/clk/CU against 128 on the
// vector unit, and the 4x4 register tile of the VALU kernel reads 8 LDS values per 16 FMAs where this reads 2 per 4096.
// =================================================================================================
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(GemmP p) {
  drop_resolve(p.drop);
  __shared__ float As[16][64 + 4];
  __shared__ float Bs[16][64 + 4];
  const int tn = (p.N + 63) / 64;
  const int tile_m = blockIdx.x / tn, tile_n = blockIdx.x - tile_m * tn;
  const int m0 = tile_m * 64, n0 = tile_n * 64;
  const int z = blockIdx.z, z0 = z % p.nb0, z1 = z / p.nb0;
  const float* A = (const float*)p.A + z0 * p.sA0 + z1 * p.sA1;
  const float* B = (const float*)p.B + z0 * p.sB0 + z1 * p.sB1;
  const long long coff = z0 * p.sC0 + z1 * p.sC1;
  const long long sar = p.transA ? 1 : p.lda, sak = p.transA ? p.lda : 1;
  const long long sbr = p.transB ? 1 : p.ldb, sbk = p.transB ? p.ldb : 1;
  int k_begin = 0, k_end = p.K;
  if (p.splitk > 1) {
    k_begin = blockIdx.y * p.ktiles_per_split * BK;
    k_end = min(p.K, k_begin + p.ktiles_per_split * BK);
    if (k_begin >= k_end) return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane & 31, lh = lane >> 5;
  const int rb = (wave >> 1) * 32, cb = (wave & 1) * 32;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = k_begin; k0 < k_end; k0 += 16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = threadIdx.x + i * 256;
      int r, k;
      if (p.transA) { r = e & 63; k = e >> 6; } else { k = e & 15; r = e >> 4; }
      float v = 0.f;
      if (m0 + r < p.M && k0 + k < k_end) v = A[(long long)(m0 + r) * sar + (long long)(k0 + k) * sak];
      As[k][r] = v;
      if (p.transB) { r = e & 63; k = e >> 6; } else { k = e & 15; r = e >> 4; }
      v = 0.f;
      if (n0 + r < p.N && k0 + k < k_end) v = B[(long long)(n0 + r) * sbr + (long long)(k0 + k) * sbk];
      Bs[k][r] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[k + lh][rb + q], Bs[k + lh][cb + q], acc, 0, 0, 0);
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + rb + (r & 3) + 8 * (r >> 2) + 4 * lh, n = n0 + cb + q;
    if (m < p.M && n < p.N) epilogue(p, z, coff, m, n, acc[r]);
  }
}

// =================================================================================================
// C ABI
// =================================================================================================
// Dispatch knobs are read by the main thread AND by the autograd thread (backward launches): environment values are
// function-local `static const` (initialised once, thread-safe since C++11), the one knob that can change at run time is atomic.
static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && e[0]) ? atoi(e) : dflt;
}
// run-time knobs (mi355x_gemm_config(key, value); first read falls back to the environment): key 4 = the 256x256 structures
// (MI355X_GEMM_V4: 0 never, 1 heuristic, 2 whenever N > 128), key 5 = the persistent structure (MI355X_GEMM_V5), key 6 = register
// prefetch instead of LDS-DMA inside the 256x256 structure (MI355X_GEMM_V6: 0 = default / 1), key 7 = the same inside the 256x128
// structure (MI355X_GEMM_V7, default 1), key 3 = fp32 problems on the matrix cores (MI355X_F32_MFMA, default 1; 0 = vector unit).  Defaults follow the in-step A/B (tools/step_ab.py, recorded graphs, same box): the
// 256x128 variant -0.2 ms per step, the 256x256 variant +0.3 ms although it wins every isolated launch (profiles/r3_gemm_structures.md)
static std::atomic<int> g_mode[10] = {{-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}};
static int mode_now(int key) {
  int v = g_mode[key].load(std::memory_order_relaxed);
  if (v < 0) {
    static const int env4 = env_int("MI355X_GEMM_V4", 1), env5 = env_int("MI355X_GEMM_V5", 1), env6 = env_int("MI355X_GEMM_V6", 0),
                     env7 = env_int("MI355X_GEMM_V7", 1), env3 = env_int("MI355X_F32_MFMA", 1), env8 = env_int("MI355X_GEMM_V8", 1),
                     env9 = env_int("MI355X_GEMM_V8_DELAY", 0);
    const int from_env = key == 4 ? env4 : key == 5 ? env5 : key == 6 ? env6 : key == 7 ? env7 : key == 3 ? env3 : key == 8 ? env8 : key == 9 ? env9 : 0;
    int expected = -1;
    g_mode[key].compare_exchange_strong(expected, from_env, std::memory_order_relaxed);
    v = g_mode[key].load(std::memory_order_relaxed);
  }
  return v;
}
static int v5_mode_now() { return mode_now(5); }
extern "C" int mi355x_gemm_config(int key, int value) {
  if (key < 3 || key > 9) return -1;
  return g_mode[key].exchange(value, std::memory_order_relaxed);
}

extern "C" int mi355x_gemm(const mi355x_gemm_desc* d, void* stream) {
  mi_clear_errors();
  if (!d || !d->A || !d->B || !d->C || d->M <= 0 || d->N <= 0 || d->K <= 0) return MI_ERR_ARG;
  if (d->in_dtype != MI_DT_F32 && d->in_dtype != MI_DT_BF16) return MI_ERR_ARG;
  // the LDS-DMA source of a reduction-major operand advances by BK * ld elements per k-tile, kept in 32 bits
  if ((d->transA && d->lda * 64 >= (1LL << 31)) || (d->transB && d->ldb * 64 >= (1LL << 31))) return MI_ERR_ARG;
  GemmP p;
  p.A = d->A; p.B = d->B; p.C = d->C;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.csc = d->c_col_stride > 0 ? d->c_col_stride : 1;
  p.transA = d->transA; p.transB = d->transB;
  p.batch = d->batch > 0 ? d->batch : 1; p.nb0 = d->nb0 > 0 ? d->nb0 : p.batch;
  p.sA0 = d->sA0; p.sA1 = d->sA1; p.sB0 = d->sB0; p.sB1 = d->sB1; p.sC0 = d->sC0; p.sC1 = d->sC1;
  p.bias = (const float*)d->bias; p.alpha = d->alpha;
  p.epi = d->epilogue; p.c_dt = d->c_dtype; p.atomic = d->atomic;
  p.swish_g = 0;
  if (p.epi == 6) { p.epi = EPI_SWISH_DROP; p.swish_g = 1; }   // MI355X_EPI_SWISH_DROP_G
  else if (p.epi == 7) { p.epi = EPI_DSWISH; p.swish_g = 1; }  // MI355X_EPI_DSWISH_G
  p.aux_in = d->aux_in; p.auxin_dt = d->aux_in_dtype; p.aux_out = d->aux_out; p.auxout_dt = d->aux_out_dtype;
  p.ldaux = d->ldaux;
  p.drop = mi_drop(d->drop_key, d->drop_threshold, d->drop_scale);
  p.colsum_out = (float*)d->colsum_out; p.colsum_stride = d->colsum_stride;
  p.g_on = 0; p.r_on = 0; p.v8_delay = 0;
  if (d->gather) {
    const mi355x_conv_gather& g = *d->gather;
    if (d->in_dtype != MI_DT_BF16 || g.C <= 0 || (g.C & 7) || g.ntaps < 1 || g.ntaps > 9 || g.nI <= 0 || g.nJ <= 0 ||
        g.SI <= 0 || g.SJ <= 0 || (g.operand != 0 && g.operand != 1))
      return MI_ERR_ARG;
    if (g.operand == 0) {  // gathered A rows (forward / dgrad)
      if (d->transA || p.batch != 1 || (g.C & 63) || d->K != g.ntaps * g.C || d->M % (g.nI * g.nJ) != 0) return MI_ERR_ARG;
    } else {               // gathered reduction-major B (weight gradient): K = positions, N = channels, batch = taps
      const long long src_elems = (long long)(d->K / (g.nI * g.nJ)) * g.SI * g.SJ * g.C;
      if (src_elems >= (1LL << 31)) return MI_ERR_ARG;
      if (!d->transA || !d->transB || p.batch != g.ntaps || p.nb0 != p.batch || d->N != g.C || d->K % (g.nI * g.nJ) != 0 ||
          ((uintptr_t)d->B & 15))
        return MI_ERR_ARG;
    }
    p.g_on = 1 + g.operand; p.g_nI = g.nI; p.g_nJ = g.nJ; p.g_SI = g.SI; p.g_SJ = g.SJ; p.g_C = g.C; p.g_si = g.si; p.g_sj = g.sj;
    p.g_ntaps = g.ntaps;
    p.g_dip = 0ull; p.g_djp = 0ull;
    for (int t = 0; t < g.ntaps; ++t) {
      if (g.di[t] < -7 || g.di[t] > 7 || g.dj[t] < -7 || g.dj[t] > 7) return MI_ERR_ARG;
      p.g_dip |= (unsigned long long)(g.di[t] + 8) << (4 * t);
      p.g_djp |= (unsigned long long)(g.dj[t] + 8) << (4 * t);
    }
  }
  if (d->rowmap) {
    const mi355x_row_map& r = *d->rowmap;
    if (r.nI <= 0 || r.nJ <= 0 || r.OI <= 0 || r.OJ <= 0 || d->M % (r.nI * r.nJ) != 0 || p.batch != 1) return MI_ERR_ARG;
    p.r_on = 1; p.r_nI = r.nI; p.r_nJ = r.nJ; p.r_OI = r.OI; p.r_OJ = r.OJ; p.r_si = r.si; p.r_sj = r.sj; p.r_oi = r.oi;
    p.r_oj = r.oj;
  }
  p.row_len = (const long long*)d->row_len; p.rows_per_b = d->rows_per_b > 0 ? d->rows_per_b : 1;
  p.rows_inner = d->rows_inner > 0 ? d->rows_inner : 1;
  if (p.epi < EPI_STORE || p.epi > EPI_MUL_POS) return MI_ERR_ARG;
  if (p.atomic && p.c_dt != MI_DT_F32) return MI_ERR_ARG;
  if ((p.epi == EPI_RESID || p.epi == EPI_DSWISH || p.epi == EPI_MUL_POS) && !p.aux_in) return MI_ERR_ARG;
  if (p.epi == EPI_SWISH_DROP && !p.aux_out) return MI_ERR_ARG;
  if (p.epi == EPI_RELU_MASK && !p.row_len) return MI_ERR_ARG;
  if (p.colsum_out && (!p.transA || d->in_dtype != MI_DT_BF16 || p.nb0 != p.batch)) return MI_ERR_ARG;
  const int nk = (p.K + BK - 1) / BK;
  int sk = d->splitk > 1 ? d->splitk : 1;
  if (sk > nk) sk = nk;
  if (sk > 1 && !p.atomic) return MI_ERR_ARG;
  p.ktiles_per_split = (nk + sk - 1) / sk;
  sk = (nk + p.ktiles_per_split - 1) / p.ktiles_per_split;
  p.splitk = sk;
  {
    // bit 0: 8-column chunks as vectors (f32 rows: two 16-byte accesses, pitch % 4 == 0; bf16 rows: one, pitch % 8 == 0);
    // bit 1: 4-column chunks as vectors (16-byte f32 / 8-byte bf16: pitch % 4 == 0) -- the last columns of a width of 4 modulo 8
    const int auxin_dt = p.epi == EPI_RESID ? MI_DT_F32 : p.auxin_dt;
    auto pitch8 = [](long long ld, int dt) { return dt == MI_DT_F32 ? !(ld & 3) : !(ld & 7); };
    bool base = p.csc == 1 && !((uintptr_t)p.C & 31) && !(p.sC0 & 7) && !(p.sC1 & 7);
    if (p.aux_in) base = base && !((uintptr_t)p.aux_in & 31);
    if (p.aux_out) base = base && !((uintptr_t)p.aux_out & 31);
    if (p.bias) base = base && !((uintptr_t)p.bias & 31);
    bool ok = base && pitch8(p.ldc, p.c_dt), ok4 = base && !(p.ldc & 3);
    if (p.aux_in) { ok = ok && pitch8(p.ldaux, auxin_dt); ok4 = ok4 && !(p.ldaux & 3); }
    if (p.aux_out) { ok = ok && pitch8(p.ldaux, p.auxout_dt); ok4 = ok4 && !(p.ldaux & 3); }
    p.vec_ok = (ok ? 1 : 0) | (ok4 ? 2 : 0);
#ifdef GEMM_ABLATE
    { const char* e = getenv("MI355X_GEMM_DBG"); if (e) p.vec_ok |= atoi(e) << 8; }
#endif
  }
  hipStream_t s = (hipStream_t)stream;
  if (d->in_dtype == MI_DT_BF16) {
    // 16-byte alignment contract of the vector loads
    if ((p.lda & 7) || (p.ldb & 7) || ((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15)) return MI_ERR_ARG;
    if ((p.sA0 & 7) || (p.sA1 & 7) || (p.sB0 & 7) || (p.sB1 & 7)) return MI_ERR_ARG;
    // K-contiguous operands are read in 8-element chunks: their pitch must cover roundup8(K) and the pad elements
    // k in [K, roundup8(K)) must be finite (zero) in memory.  Reduction-major operands need pitch >= roundup8(rows):
    // a partial chunk's extra columns only feed output rows/cols >= M/N, which are never stored.
    const int K8 = (p.K + 7) & ~7;
    if (!p.transA && !p.g_on && p.lda < K8) return MI_ERR_ARG;
    if (!p.transB && p.ldb < K8) return MI_ERR_ARG;
    if (p.transA && p.lda < ((p.M + 7) & ~7)) return MI_ERR_ARG;
    if (p.transB && p.g_on != 2 && p.ldb < ((p.N + 7) & ~7)) return MI_ERR_ARG;
    const int tm = (p.M + BM - 1) / BM, tn = (p.N + BN - 1) / BN;
    dim3 grid(tm * tn, sk, p.batch);
    static const int use_v2 = env_int("MI355X_GEMM_V2", 1) ? 1 : 0;
    if (p.g_on && !(use_v2 && p.M >= 192 && p.N >= 96)) return MI_ERR_ARG;  // the gather lives in the LDS-DMA structures
    // few output tiles (e.g. M = 8032 rows x N = 512: 128 tiles of 256x128 on 256 CUs): the 128x128 structure doubles the
    // workgroups and wins in isolation although its K loop is slower (FFN2 forward at M = 8032: 46.3 -> 39.5 us); inside a
    // training step, next to the weight-gradient stream, it only paid off below ~100 tiles (Squeezeformer-Medium's N = 324
    // launches at the reduced frame rate: step 47.95 -> 46.95 ms; FastConformer's 128-tile launches: 32.47 -> 32.67 ms)
    // Round 6: with the vector tail / templated partial-tile epilogue for widths of 4 modulo 8 the 256x128 structure is ahead again
    // on those launches too (Squeezeformer-Medium 35.68 -> 35.52 ms, same box, twice; Transducer unchanged): default 0 = rule off.
    static const int few_mode = env_int("MI355X_GEMM_FEW_TILES", 0);
    const long long blocks256 = (long long)((p.M + BM2 - 1) / BM2) * tn * sk * p.batch;
    const bool few_tiles = !p.g_on && !p.r_on && !p.atomic && blocks256 <= few_mode && p.N <= 1024 &&
                           (long long)tm * tn * sk * p.batch > blocks256;
    if (use_v2 && p.M >= 192 && p.N >= 96 && !(p.transA && !p.transB) && !few_tiles) {
      const int tm2 = (p.M + BM2 - 1) / BM2;
      const int shm = 3 * NT2_STAGE * 2;
      static const bool attr_ok = [shm] {
        bool ok = hipFuncSetAttribute((const void*)gemm_bf16_v2_kernel<false, false>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, shm) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void*)gemm_bf16_v2_kernel<false, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, shm) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void*)gemm_bf16_v2_kernel<true, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, shm) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void*)gemm_bf16_v2_kernel<false, false, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, shm) == hipSuccess;
        return ok;
      }();
      if (!attr_ok) { (void)hipGetLastError(); return MI_ERR_LAUNCH; }
      dim3 grid2(tm2 * tn, sk, p.batch);
      // phase-staggered 256x256 structure on 16x16x32 MFMAs (key 8 / MI355X_GEMM_V8: 0 = never, 1 = where it measured faster, 2 = every
      // shape it can run; 3 = as 1 plus the weight-gradient (TN) layouts, 4 = as 1 plus the wide plain stores the persistent
      // structure otherwise takes, 5 = as 1 plus the 128x256 tile where 256x256 tiles do not fill the chip and K >= 768 -- A/B arms).  NT: K-contiguous operands, whole K-tiles; TN: both operands reduction-major (K tail
      // allowed); at least two K-tiles per workgroup, 32-bit operand offsets.
      // The TN layouts are correct on it but SLOWER than on the lock-step structures (conv2 weight gradient 2.54 vs 2.02 ms, a layer's
      // grouped weight gradients 307 vs 286 us, profiles/r6_gemm_8phase.md): a fragment is two ds_read_b64_tr_b16, and the 8-byte
      // reads only reach the LDS rate with both waves of a SIMD reading -- the stagger has one of them reading at a time.
      {
        const int v8_mode = mode_now(8);
        const int nk_wg8 = sk > 1 ? p.ktiles_per_split : nk;
        const bool nt8 = !p.transA && !p.transB, tn8l = p.transA && p.transB;
        const int last8 = nk - (sk - 1) * nk_wg8;  // K-tiles of the last slice
        bool v8_can = (nt8 || tn8l) && nk_wg8 >= 2 && last8 >= 2 && p.N > 128;
        if (nt8) {
          v8_can = v8_can && p.g_on != 2 && !(p.K % BK) && (long long)p.N * p.ldb < (1LL << 31);
          if (p.g_on == 1)  // gathered A: 32-bit byte offsets into the source grid, a K-tile inside one tap
            v8_can = v8_can && !(p.g_C % BK) && (long long)(p.M / (p.g_nI * p.g_nJ)) * p.g_SI * p.g_SJ * p.g_C < (1LL << 31);
          else v8_can = v8_can && (long long)p.M * p.lda < (1LL << 31);
        } else if (tn8l) {
          v8_can = v8_can && p.g_on != 1 && 64 * p.lda < (1LL << 30) && (p.g_on == 2 || 64 * p.ldb < (1LL << 30));
        }
        const int tn8 = (p.N + BN4 - 1) / BN4;
        const long long blocks8 = (long long)tm2 * tn8 * sk * p.batch;
        const long long blocks2_ = (long long)tm2 * tn * sk * p.batch;
        const double eff8 = (double)blocks8 / (double)(((blocks8 + 255) / 256) * 256);
        const double eff2_ = (double)blocks2_ / (double)(((blocks2_ + 255) / 256) * 256);
        // the rule of the third structure: the larger tile still fills the chip, few padded columns, no worse wave quantisation
        const bool fills = blocks8 >= 224 && (long long)tn8 * BN4 * 8 <= (long long)p.N * 9 && eff8 >= 0.9 * eff2_;
        // (plain stores at least 1536 columns wide with K <= 576 stay on the persistent structure where the 256x256 tiles quantise
        //  badly: 40.3 vs 41.4 us on the QKV shape, 378 tiles = 1.48 rounds; N = 2048 -- 504 tiles -- is 44.7 vs 53.6 us the other way)
        const bool v5_keeps = nt8 && !p.g_on && p.epi == EPI_STORE && p.N >= 1536 && nk >= 8 && nk <= 9 && !p.atomic && sk == 1 &&
                              !(p.N % BN) && v5_mode_now() && v8_mode != 4 && eff8 < 0.85;
        const bool v8_pick = v8_mode == 2 || (v8_mode >= 1 && fills && !v5_keeps && (nt8 || v8_mode == 3));
        // the 128x256 tile of the same structure (modes 2 and 5): dense NT problems whose 256x256 tiles would leave the chip
        // half empty but whose 128x256 tiles fill it (N = 512 at M = 16032: 126 -> 252 workgroups)
        {
          const long long blocks1 = (long long)((p.M + 127) / 128) * tn8 * sk * p.batch;
          const double eff1 = (double)blocks1 / (double)(((blocks1 + 255) / 256) * 256);
          const bool fills1 = blocks1 >= 224 && (long long)tn8 * BN4 * 8 <= (long long)p.N * 9 && eff1 >= 0.9 * eff2_;
          // (K >= 768: at K = 512 -- the 512 x 512 projections -- the persistent / 256x128 structures are ahead, 23.6 vs 24.6 us)
          // Isolated it wins 3-6 % over the 256x128 lock-step structure (FFN2 forward 36.2 -> 34.1 us); INSIDE the training step it
          // loses 0.16 ms (36.50 vs 36.66 ms, same box, interleaved) -- not in the default set.
          if (v8_can && nt8 && !p.g_on && !fills && fills1 && ((v8_mode == 5 && nk >= 12) || (v8_mode == 2 && blocks8 < 224))) {
            static const bool attr81_ok = hipFuncSetAttribute((const void*)gemm_bf16_v8_kernel<0, false, 1>,
                                                              hipFuncAttributeMaxDynamicSharedMemorySize, 9 * V8_HALF_B) == hipSuccess;
            if (!attr81_ok) { (void)hipGetLastError(); return MI_ERR_LAUNCH; }
            MI_LAUNCH((gemm_bf16_v8_kernel<0, false, 1>), dim3(((p.M + 127) / 128) * tn8, sk, p.batch), dim3(512), 9 * V8_HALF_B, s, p);
            return mi_check_launch();
          }
        }
        if (v8_can && v8_pick) {
          typedef void (*v8_fn)(GemmP);
          static const v8_fn v8_all[] = {gemm_bf16_v8_kernel<0, false>, gemm_bf16_v8_kernel<1, false>, gemm_bf16_v8_kernel<0, true>,
                                         gemm_bf16_v8_kernel<2, true>};
          static const bool attr8_ok = [] {
            for (v8_fn f : v8_all)
              if (hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * V8_BUF_B) != hipSuccess) return false;
            return true;
          }();
          if (!attr8_ok) { (void)hipGetLastError(); return MI_ERR_LAUNCH; }
          const v8_fn fn = nt8 ? v8_all[p.g_on == 1 ? 1 : 0] : v8_all[p.g_on == 2 ? 3 : 2];
          // key 9: the phase offset (10-ns ticks; > 0: that many for every problem of more than one round and at most 16 K-tiles)
          if (nt8 && blocks8 > 256 && nk <= 16) p.v8_delay = mode_now(9);
          MI_LAUNCH(fn, dim3(tm2 * tn8, sk, p.batch), dim3(512), 2 * V8_BUF_B, s, p);
          return mi_check_launch();
        }
      }
      // persistent 256x128 structure with the epilogue overlapped into the next tile's K loop: the Conformer block's
      // forward / dgrad GEMMs (dense NT, full-width vector epilogue, K >= 8 K-tiles, at least one tile per CU)
      const int v5_mode = v5_mode_now();
      // v5_mode 1: where it measured faster than the tiled structures (K <= 576: Swish-gradient, residual, and stores at
      // least 1536 columns wide); 2: every shape it can run (tests, A/B)
      const bool v5_epi = (p.epi == EPI_STORE && (v5_mode == 2 || p.N >= 1536)) || (p.epi == EPI_SWISH_DROP && v5_mode == 2) ||
                          (p.epi == EPI_RESID && p.c_dt == MI_DT_F32) || (p.epi == EPI_DSWISH && p.auxin_dt == MI_DT_BF16);
      if (v5_mode && !p.transA && !p.transB && !p.g_on && !p.r_on && !p.atomic && p.batch == 1 && sk == 1 && (p.vec_ok & 1) &&
          !(p.N % BN) && nk >= 8 && (nk <= 9 || v5_mode == 2) && tm2 * tn >= 256 && v5_epi && (long long)p.M * p.lda < (1LL << 31) &&
          (long long)p.N * p.ldb < (1LL << 31) && (long long)p.M * p.ldc < (1LL << 31) &&
          (long long)p.M * p.ldaux < (1LL << 31)) {
        static const bool attr5_ok = [] {
          bool ok = hipFuncSetAttribute((const void*)gemm_bf16_v5_kernel<EPI_STORE>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, V5_LDS_BYTES) == hipSuccess;
          ok = ok && hipFuncSetAttribute((const void*)gemm_bf16_v5_kernel<EPI_SWISH_DROP>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, V5_LDS_BYTES) == hipSuccess;
          ok = ok && hipFuncSetAttribute((const void*)gemm_bf16_v5_kernel<EPI_RESID>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, V5_LDS_BYTES) == hipSuccess;
          ok = ok && hipFuncSetAttribute((const void*)gemm_bf16_v5_kernel<EPI_DSWISH>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, V5_LDS_BYTES) == hipSuccess;
          return ok;
        }();
        if (!attr5_ok) { (void)hipGetLastError(); return MI_ERR_LAUNCH; }
        const int nt5 = tm2 * tn;
        dim3 grid5(256);
        switch (p.epi) {
          case EPI_STORE: MI_LAUNCH((gemm_bf16_v5_kernel<EPI_STORE>), grid5, dim3(512), V5_LDS_BYTES, s, p, nt5, tn); break;
          case EPI_SWISH_DROP: MI_LAUNCH((gemm_bf16_v5_kernel<EPI_SWISH_DROP>), grid5, dim3(512), V5_LDS_BYTES, s, p, nt5, tn); break;
          case EPI_RESID: MI_LAUNCH((gemm_bf16_v5_kernel<EPI_RESID>), grid5, dim3(512), V5_LDS_BYTES, s, p, nt5, tn); break;
          default: MI_LAUNCH((gemm_bf16_v5_kernel<EPI_DSWISH>), grid5, dim3(512), V5_LDS_BYTES, s, p, nt5, tn); break;
        }
        return mi_check_launch();
      }
      // 256x256 structure when the problem still fills the chip with the larger tile
      const int v4_mode = mode_now(4);  // 0 = never, 1 = heuristic (default), 2 = whenever N >= 129
      const int tn4 = (p.N + BN4 - 1) / BN4;
      const long long blocks4 = (long long)tm2 * tn4 * sk * p.batch;
      const bool waste_ok = (long long)tn4 * BN4 * 8 <= (long long)p.N * 9;  // <= 12.5 % padded columns
      // wave quantisation: workgroups / (rounds x 256 CUs) must not fall behind the 256x128 tiling by more than 10 %
      const long long blocks2 = (long long)tm2 * tn * sk * p.batch;
      const double eff4 = (double)blocks4 / (double)(((blocks4 + 255) / 256) * 256);
      const double eff2 = (double)blocks2 / (double)(((blocks2 + 255) / 256) * 256);
      if ((v4_mode == 2 && p.N > 128) || (v4_mode == 1 && blocks4 >= 224 && waste_ok && eff4 >= 0.9 * eff2)) {
        const int shm4 = 2 * NT4_STAGE * 2;
        // register-prefetch structure (sixth): dense K-contiguous operands, whole K-tiles, 32-bit operand offsets
        const int v6_mode = mode_now(6);
        if (v6_mode && !p.transA && !p.transB && !p.g_on && sk == 1 && !(p.K % (2 * BK)) && p.K >= 4 * BK && !(p.lda & 7) && !(p.ldb & 7) &&
            !((uintptr_t)p.A & 15) && !((uintptr_t)p.B & 15) && (long long)p.M * p.lda < (1LL << 30) &&
            (long long)p.N * p.ldb < (1LL << 30)) {
          static const bool attr6_ok = hipFuncSetAttribute((const void*)gemm_bf16_v6_kernel<0>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, shm4) == hipSuccess &&
                                       hipFuncSetAttribute((const void*)gemm_bf16_v6_kernel<1>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, shm4) == hipSuccess;
          if (!attr6_ok) { (void)hipGetLastError(); return MI_ERR_LAUNCH; }
          if (v6_mode == 2) MI_LAUNCH((gemm_bf16_v6_kernel<0>), dim3(tm2 * tn4, 1, p.batch), dim3(512), shm4, s, p);
          else MI_LAUNCH((gemm_bf16_v6_kernel<1>), dim3(tm2 * tn4, 1, p.batch), dim3(512), shm4, s, p);
          return mi_check_launch();
        }
        typedef void (*v4_fn)(GemmP);
        static const v4_fn v4_all[] = {gemm_bf16_v4_kernel<false, false, 0>, gemm_bf16_v4_kernel<false, false, 1>,
                                       gemm_bf16_v4_kernel<false, true, 0>, gemm_bf16_v4_kernel<true, true, 0>,
                                       gemm_bf16_v4_kernel<true, true, 2>};
        static const bool attr4_ok = [shm4] {
          for (v4_fn f : v4_all)
            if (hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, shm4) != hipSuccess) return false;
          return true;
        }();
        if (!attr4_ok) { (void)hipGetLastError(); return MI_ERR_LAUNCH; }
        dim3 grid4(tm2 * tn4, sk, p.batch);
        v4_fn fn;
        if (!p.transA && !p.transB) fn = v4_all[p.g_on == 1 ? 1 : 0];
        else if (!p.transA && p.transB) fn = v4_all[2];
        else fn = v4_all[p.g_on == 2 ? 4 : 3];
        MI_LAUNCH(fn, grid4, dim3(512), shm4, s, p);
      } else
      if (!p.transA && !p.transB) {
        // register-prefetch K loop (key 7): dense K-contiguous operands, an even number (>= 4) of whole K-tiles per workgroup
        const int nk_wg = sk > 1 ? p.ktiles_per_split : nk;
        const bool reg_ok = mode_now(7) && !p.g_on && !(p.K % BK) && nk_wg >= 4 && !(nk_wg & 1) && (sk == 1 || !(nk % nk_wg)) &&
                            !(p.lda & 7) && !(p.ldb & 7) && !((uintptr_t)p.A & 15) && !((uintptr_t)p.B & 15) &&
                            (long long)p.M * p.lda < (1LL << 30) && (long long)p.N * p.ldb < (1LL << 30);
        if (reg_ok) MI_LAUNCH((gemm_bf16_v2_kernel<false, false, true>), grid2, dim3(512), shm, s, p);
        else MI_LAUNCH((gemm_bf16_v2_kernel<false, false>), grid2, dim3(512), shm, s, p);
      } else if (!p.transA && p.transB) MI_LAUNCH((gemm_bf16_v2_kernel<false, true>), grid2, dim3(512), shm, s, p);
      else MI_LAUNCH((gemm_bf16_v2_kernel<true, true>), grid2, dim3(512), shm, s, p);
    } else
    if (!p.transA && !p.transB) MI_LAUNCH((gemm_bf16_kernel<false, false>), grid, dim3(256), 0, s, p);
    else if (!p.transA && p.transB) MI_LAUNCH((gemm_bf16_kernel<false, true>), grid, dim3(256), 0, s, p);
    else if (p.transA && p.transB) MI_LAUNCH((gemm_bf16_kernel<true, true>), grid, dim3(256), 0, s, p);
    else MI_LAUNCH((gemm_bf16_kernel<true, false>), grid, dim3(256), 0, s, p);
  } else {
    const int tm = (p.M + 63) / 64, tn = (p.N + 63) / 64;
    dim3 grid(tm * tn, sk, p.batch);
    // key 3 / MI355X_F32_MFMA=0 keeps the vector-unit kernel (A/B, and the reference point of tests/test_kernels_gpu.py)
    if (mode_now(3)) MI_LAUNCH(gemm_f32_mfma_kernel, grid, dim3(256), 0, s, p);
    else MI_LAUNCH(gemm_f32_kernel, grid, dim3(256), 0, s, p);
  }
  return mi_check_launch();
}

extern "C" int mi355x_gemm_grouped(const mi355x_gemm_desc* descs, int n, void* stream) {
  mi_clear_errors();
  if (!descs || n < 1 || n > GRP_MAX) return MI_ERR_ARG;
  GroupP g;
  g.n = n; g.K = descs[0].K;
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    const mi355x_gemm_desc& d = descs[i];
    if (!d.A || !d.B || !d.C || d.M <= 0 || d.N <= 0 || d.K != g.K || d.K <= 0) return MI_ERR_ARG;
    if (d.in_dtype != MI_DT_BF16 || d.c_dtype != MI_DT_F32 || !d.transA || !d.transB || !d.atomic) return MI_ERR_ARG;
    if (d.batch > 1 || d.bias || d.aux_in || d.aux_out || d.gather || d.rowmap || d.epilogue != EPI_STORE) return MI_ERR_ARG;
    if (d.alpha != 1.f || d.c_col_stride > 1 || d.drop_threshold != 0u) return MI_ERR_ARG;
    if ((d.lda & 7) || (d.ldb & 7) || ((uintptr_t)d.A & 15) || ((uintptr_t)d.B & 15)) return MI_ERR_ARG;
    if (d.lda < ((d.M + 7) & ~7) || d.ldb < ((d.N + 7) & ~7)) return MI_ERR_ARG;
    g.A[i] = d.A; g.B[i] = d.B; g.C[i] = d.C; g.colsum[i] = (float*)d.colsum_out;
    g.M[i] = d.M; g.N[i] = d.N; g.lda[i] = d.lda; g.ldb[i] = d.ldb; g.ldc[i] = d.ldc;
    g.tile_begin[i] = tiles;
    tiles += ((d.M + BM2 - 1) / BM2) * ((d.N + BN - 1) / BN);
  }
  for (int i = n; i <= GRP_MAX; ++i) g.tile_begin[i] = tiles;
  const int nk = (g.K + BK - 1) / BK;
  int sk = descs[0].splitk > 1 ? descs[0].splitk : 1;
  if (sk > nk) sk = nk;
  // eighth structure (key 8 modes 2 and 3 only: slower than this one on the reduction-major layouts, see mi355x_gemm): 256x256 tiles --
  // half as many as the caller's split-K factor was chosen for, so the factor is chosen again here by the same rule (whole rounds
  // of the 256 CUs, smaller factors preferred, >= 16 K-tiles a slice)
  const int v8_mode = mode_now(8);
  if ((v8_mode == 2 || v8_mode == 3) && nk >= 4) {
    bool ok8 = true;
    int tiles8 = 0;
    for (int i = 0; i < n; ++i) {
      ok8 = ok8 && descs[i].N > 128 && 64 * descs[i].lda < (1LL << 30) && 64 * descs[i].ldb < (1LL << 30);
      g.tile_begin[i] = tiles8;
      tiles8 += ((descs[i].M + BM2 - 1) / BM2) * ((descs[i].N + BN4 - 1) / BN4);
    }
    if (ok8) {
      for (int i = n; i <= GRP_MAX; ++i) g.tile_begin[i] = tiles8;
      int best = 1;
      double best_score = -1.0;
      for (int c = 1; c <= 16; ++c) {
        if (c > 1 && nk / c < 16) break;
        const long long blocks = (long long)tiles8 * c;
        const double score = (double)blocks / (double)(((blocks + 255) / 256) * 256) - 0.01 * c;
        if (score > best_score + 1e-9) { best = c; best_score = score; }
      }
      if (sk > 1) sk = best;
      g.ktiles_per_split = (nk + sk - 1) / sk;
      if (g.ktiles_per_split < 2) { g.ktiles_per_split = 2; }
      sk = (nk + g.ktiles_per_split - 1) / g.ktiles_per_split;
      if (nk - (sk - 1) * g.ktiles_per_split >= 2) {  // (every slice needs two K-tiles)
        g.splitk = sk;
        static const bool attr8g_ok = hipFuncSetAttribute((const void*)gemm_bf16_grouped_tn8_kernel,
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, 2 * V8_BUF_B) == hipSuccess;
        if (!attr8g_ok) { (void)hipGetLastError(); return MI_ERR_LAUNCH; }
        MI_LAUNCH(gemm_bf16_grouped_tn8_kernel, dim3(tiles8, sk, 1), dim3(512), 2 * V8_BUF_B, (hipStream_t)stream, g);
        return mi_check_launch();
      }
    }
    // not taken: the tile list of the 256x128 structure again
    tiles = 0;
    for (int i = 0; i < n; ++i) {
      g.tile_begin[i] = tiles;
      tiles += ((descs[i].M + BM2 - 1) / BM2) * ((descs[i].N + BN - 1) / BN);
    }
    for (int i = n; i <= GRP_MAX; ++i) g.tile_begin[i] = tiles;
    sk = descs[0].splitk > 1 ? descs[0].splitk : 1;
    if (sk > nk) sk = nk;
  }
  g.ktiles_per_split = (nk + sk - 1) / sk;
  sk = (nk + g.ktiles_per_split - 1) / g.ktiles_per_split;
  g.splitk = sk;
  const int shm = 3 * NT2_STAGE * 2;
  static const bool attr_ok = hipFuncSetAttribute((const void*)gemm_bf16_grouped_tn_kernel,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, shm) == hipSuccess;
  if (!attr_ok) { (void)hipGetLastError(); return MI_ERR_LAUNCH; }
  MI_LAUNCH(gemm_bf16_grouped_tn_kernel, dim3(tiles, sk, 1), dim3(512), shm, (hipStream_t)stream, g);
  return mi_check_launch();
}

