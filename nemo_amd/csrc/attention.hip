// Fused relative-position multi-head self-attention for MI355X (bf16, d_k = 64), flash-style: the [B,H,T,T] score matrix
// and the [B,H,T,2T-1] positional matrix of the reference are never materialised in HBM.
//
//   score[i,j] = ((q_i + u_h) . k_j + (q_i + v_h) . p_{T-1+j-i}) / sqrt(d_k)   (rel_shift == the index map c = T-1+j-i)
//   masked keys / queries (>= len[b]) excluded, softmax, dropout, context = P @ V
//
// Every product keeps "lane = query": S^T = K Qu^T, G^T = P_band Qv^T, O^T = V^T P^T (mfma_f32_32x32x16_bf16, C/D layout
// col = lane&31 = query, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)).  Consequences:
//   * row max / row sum of the online softmax are an in-lane reduction over 16 registers + one xor-32 shuffle;
//   * the running rescale alpha is a per-lane scalar applied to the lane's own O^T registers;
//   * the rel_shift becomes a PER-LANE CONSTANT row offset into G^T (c - c_min = key_row + 31 - q), resolved through a
//     wave-private LDS tile [32 queries][66] (stride 65 between lanes: conflict-free reads);
//   * P^T is consumed as the MFMA B operand straight from the S^T accumulator registers (the k-slot <-> key mapping of
//     an MFMA is arbitrary as long as A and B agree), V^T fragments come from ds_read_b64_tr_b16 on the [key][dv] tile.
// K / V tiles (double-buffered) and the positional band (a ring of six 32-row blocks: consecutive key steps share four of
// their five blocks, a step stages one new block) are staged one step ahead by inline-asm LDS-DMA (dma16) behind a single
// barrier per step.  73-77 KiB LDS -> two workgroups per CU.
//
// Replaces on the reference path: RelPositionMultiHeadAttention.forward + MultiHeadAttention.forward_attention
//   (nemo/collections/asr/parts/submodules/multi_head_attention.py:272-354, 124-146): two batched matmuls, pad/view/slice
//   rel_shift, two masked_fill, softmax, dropout, matmul.
#include <stdlib.h>
#include "common.h"
#include "mi355x_asr.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

typedef __attribute__((address_space(3))) const char lds_cchar_t;
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(lds_cchar_t*)p; }

// LDS-DMA by inline asm.  Behind a `__builtin_amdgcn_global_load_lds` hipcc puts `s_waitcnt vmcnt(0)` in front of LDS reads
// it cannot tell apart from the DMA's target (transpose reads, merged reads, ...): a prefetch issued at the top of a step was
// waited for at the step's first such read.  Issued by hand the compiler knows nothing about the transfer; every consumer in
// this file sits behind an explicit `s_waitcnt vmcnt` + barrier anyway.  lds_base must be wave-uniform (lane i lands at
// lds_base + 16 i).  M0 is not used by anything else in this translation unit.
__device__ __forceinline__ void dma16(const void* src, const void* lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
               :
               : "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_addr(lds_base)))
               : "memory");
}

__device__ __forceinline__ void dma4(const void* src, const void* lds_base) {  // lane i lands at lds_base + 4 i
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off"
               :
               : "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_addr(lds_base)))
               : "memory");
}

// head width: a template parameter of every kernel (ADK = 64: the tuned form, two workgroups per CU; ADK = 128 -- round 5, heads of
// 65..128 lanes zero-padded to 128, Squeezeformer-Medium's d_k = 81 -- the same code with 8 k-steps, 4 output tiles and one
// workgroup per CU).  Rows of every staged tile are ADK elements = ADK / 8 sixteen-byte chunks.
#define ABQ 128          // queries per workgroup (4 waves x 32)
#define ABK 32           // keys per step
#define ABAND (ABQ + ABK)  // positional band rows staged per step (159 needed)
#define SG_LD 66
#define APRING 6         // positional-band ring: blocks of 32 rows (5 in use + 1 being staged)

// Dropout of attention probabilities: keep(b,h,i,j) from full-rate integer ops only (v_mul_u32_u24, shifts, xors) so that
// it costs the same whether a lane walks keys (forward / dQ: lane = query) or queries (dK/dV: lane = key).  `akey` is a
// strong hash of (seed, site, b, h) computed once per workgroup.
__device__ __forceinline__ uint32_t attn_key(const DropCfg& d, int b, int h) {
  return mix32(mix32(d.key ^ (uint32_t)(b * 0x632BE5AB)) + (uint32_t)h * 0x9E3779B9u);
}
// Dropout on the attention probabilities, counter-based and two-level; the three loops are VALU-bound, so the definition is
// shaped by what it costs per element in BOTH register layouts ("lane = query" walking keys: forward / dQ; "lane = key"
// walking queries: dK/dV) -- 5.5 VALU ops per element where the first version needed 22 and the second 10.75:
//   * every 4x4 block (bi = i>>2, bj = j>>2) of the [T,T] matrix gets a seed = one multiply-add round over R(bi) ^ C(bj).
//     R and C are full mix32 hashes: the one that is constant per lane is hoisted out of the loop, the one that changes per
//     step is wave-uniform up to the lane-half bit, so both candidates are computed on the SCALAR unit and selected (1 op);
//   * a pair of horizontally adjacent elements (i, j&~1), (i, j|1) shares y = t ^ (t >> 15), t = seed + ((i&3)*2 + ((j&3)>>1)) * G;
//     the even column keeps iff mul24(y, KA) >= threshold, the odd one iff mul24(y, KB) >= threshold (two different odd 24-bit
//     multipliers: exhaustively over the 2^24 values of y the two decisions are independent to 1e-5, tools/attn_drop_stats.py).
//     "Lane = query" layouts need both decisions of a pair (5 ops per 2 elements), "lane = key" layouts one of them with the
//     multiplier chosen per lane (4 ops per element);
//   * the 1/(1-p) scale is not applied per element: forward folds it into the final normalisation, the backward kernels into
//     constants / the accumulators.
// attn_drop() is the scalar definition.
#define ADROP_G 0x9E3779B9u
#define ADROP_KA 0x2C1B3Du
#define ADROP_KB 0x5A2D39u
__device__ __forceinline__ uint32_t adrop_rcode(uint32_t akey, uint32_t bi) { return mix32(akey + bi * 0x9E3779B9u); }
__device__ __forceinline__ uint32_t adrop_ccode(uint32_t bj) { return mix32(bj * 0x85EBCA6Bu + 0x165667B1u); }
__device__ __forceinline__ uint32_t adrop_seed(uint32_t x) { return __umul24(x, 0x846CA7u) + (x >> 13); }  // v_mad_u32_u24
__device__ __forceinline__ uint32_t adrop_y(uint32_t t) { return t ^ (t >> 15); }
__device__ __forceinline__ bool attn_keep(const DropCfg& d, uint32_t akey, int i, int j) {
  if (d.threshold == 0u) return true;
  const uint32_t seed = adrop_seed(adrop_rcode(akey, (uint32_t)i >> 2) ^ adrop_ccode((uint32_t)j >> 2));
  const uint32_t y = adrop_y(seed + (uint32_t)((i & 3) * 2 + ((j & 3) >> 1)) * ADROP_G);
  return __umul24(y, (j & 1) ? ADROP_KB : ADROP_KA) >= d.threshold;
}
__device__ __forceinline__ float attn_drop(const DropCfg& d, uint32_t akey, int i, int j) {
  return attn_keep(d, akey, i, j) ? d.scale : 0.f;
}

template <int ADK>
__device__ __forceinline__ int a_off(int r, int chunk) { return r * ADK + ((chunk ^ ((r >> 1) & 7)) << 3); }

// rows [row0, row0+nrows) x 64 columns (col offset applied by caller) -> LDS image [nrows][64] with the (row>>1)&7 swizzle
// applied on the source side; rows are clamped into [0, rmax]; nchunks = rows * (ADK / 8)
template <int ADK>
__device__ __forceinline__ void stage_rows(const bf16_t* base, long long ld, int row0, int rmax, bf16_t* lds, int nchunks) {
  constexpr int NCH = ADK / 8;
  const int wave = threadIdx.x >> 6;
  for (int q0 = 0; q0 < nchunks; q0 += 256) {
    const int q = q0 + threadIdx.x;
    if (q0 + wave * 64 < nchunks) {  // wave-uniform
      const int r = q / NCH, ck = q % NCH;
      const int gck = ck ^ ((r >> 1) & 7);
      int gr = row0 + r;
      gr = gr < 0 ? 0 : (gr > rmax ? rmax : gr);
      // rows and pitches are far below 2^24 (the batch offset is in `base`): one full-rate 24-bit multiply instead of the
      // 64-bit sequence (2 quarter-rate v_mul_lo_u32 + v_mad_u64_u32) per address
      const bf16_t* src = base + __mul24(gr, (int)ld) + gck * 8;
      bf16_t* dst = lds + (q0 + wave * 64) * 8;
      dma16(src, dst);
    }
  }
}
// V tile [32 keys][ADK dv] kept exactly as in memory (for the transpose read), chunk swizzle c ^= (k&3)<<1
template <int ADK>
__device__ __forceinline__ void stage_v(const bf16_t* base, long long ld, int row0, int rmax, bf16_t* lds) {
  constexpr int NCH = ADK / 8;
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int q0 = 0; q0 < 32 * NCH; q0 += 256) {  // 256 chunks per pass
    const int q = q0 + threadIdx.x;
    const int k = q / NCH, cp = q % NCH;
    const int c = cp ^ ((k & 3) << 1);
    int gr = row0 + k;
    gr = gr > rmax ? rmax : gr;
    const bf16_t* src = base + __mul24(gr, (int)ld) + c * 8;
    bf16_t* dst = lds + (q0 + wave * 64) * 8;
    dma16(src, dst);
  }
}
// A fragment of V^T for the P.V product: lane (dv = dv0 + (lane&31)) gets V[key slots of (s, half)][dv]
template <int ADK>
__device__ __forceinline__ bf16x8 vt_frag(const bf16_t* vt, int dv0, int s, int lane) {
  const int t = lane & 15, g4 = (lane >> 4) & 1, lh = lane >> 5;
  const int col = dv0 + g4 * 16 + (t & 3) * 4;
  union { bf16x8 v; s16x4 h[2]; } u;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int krow = 16 * s + 8 * r + 4 * lh + (t >> 2);  // keys (4*lh + 0..3) and (8 + 4*lh + 0..3) of K16 step s
    const int off = krow * ADK + (((col >> 3) ^ ((krow & 3) << 1)) << 3) + (col & 7);
    u.h[r] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(vt + off));
  }
  return u.v;
}

__device__ __forceinline__ bf16x8 pack8(const float* v) {
  union { bf16x8 v; uint32_t w[4]; } u;
#pragma unroll
  for (int j = 0; j < 4; ++j) u.w[j] = pack_bf2(v[2 * j], v[2 * j + 1]);
  return u.v;
}

// q fragments (B operand): lane holds (q + bias)[query = lane&31][dk = kk*16 + (lane>>5)*8 + e]
template <int NKK>
__device__ __forceinline__ void load_q(const bf16_t* qrow, const float* bias, bf16x8 (&out)[NKK], bool valid, int lh) {
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) {
    float v[8];
    const int k0 = kk * 16 + lh * 8;
    if (valid) {
      const u32x4 t = *reinterpret_cast<const u32x4*>(qrow + k0);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(t[j] << 16); v[2 * j + 1] = __uint_as_float(t[j] & 0xffff0000u); }
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += bias[k0 + j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
    out[kk] = pack8(v);
  }
}

template <int ADK>
__global__ __launch_bounds__(256, ADK == 64 ? 2 : 1) void relpos_flash_fwd_kernel(const bf16_t* __restrict__ qkv, long long ldq,
                                                                  const bf16_t* __restrict__ pos, long long ldp,
                                                                  const float* __restrict__ bias_u, const float* __restrict__ bias_v,
                                                                  const long long* __restrict__ len, bf16_t* __restrict__ ctx,
                                                                  bf16_t* __restrict__ ctx_lo, long long ldo,
                                                                  float* __restrict__ lse, int B, int H, int T,
                                                                  int Tp, float scale, DropCfg drop,
                                                                  const long long* __restrict__ cu) {
  drop_resolve(drop);
  constexpr int NKK = ADK / 16, NDT = ADK / 32, NCH = ADK / 8;
  __shared__ __attribute__((aligned(16))) bf16_t s_k2[2][ABK * ADK];   // 2 x 4 KiB (double-buffered)
  __shared__ __attribute__((aligned(16))) bf16_t s_v2[2][ABK * ADK];   // 2 x 4 KiB
  __shared__ __attribute__((aligned(16))) bf16_t s_p[APRING * 32 * ADK];  // 24 KiB: ring of 6 blocks of 32 band rows
  __shared__ __attribute__((aligned(16))) float s_g[4][32 * SG_LD];    // 33 KiB

  const int b = blockIdx.z, h = blockIdx.y;
  const int i0_blk = blockIdx.x * ABQ;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane & 31, lh = lane >> 5;
  const int i = i0_blk + wave * 32 + q;  // this lane's query
  const int L = (int)min((long long)T, len[b]);
  const bool qvalid = i < L;
  const int P = 2 * T - 1;
  // PACKED rows (cu != nullptr): the activation matrices hold only the valid frames of every utterance, utterance b at
  // rows cu[b] .. cu[b] + L - 1; T stays the padded length (it fixes the positional geometry and the [B, H, T] statistics).
  const long long row0 = cu ? cu[b] : (long long)b * T;
  const int Tr = cu ? L : T;                  // rows of this utterance that exist
  const int Trc = Tr > 0 ? Tr - 1 : 0;        // the row index loads are clamped to

  if (cu && i0_blk >= L) {
    // packed rows: a query tile beyond the utterance has no row to write; only its (never used) statistics are cleared.  On the
    // padded grid the same tile still runs: its rows exist and must hold finite numbers (the weight gradients sum over every row).
    if (lse && lh == 0 && i < T) lse[((long long)b * H + h) * T + i] = 0.f;
    return;
  }
  const bf16_t* qbase = qkv + row0 * ldq + h * ADK;
  const bf16_t* kbase = qbase + (ldq / 3);
  const bf16_t* vbase = qbase + 2 * (ldq / 3);
  const bf16_t* pbase = pos + h * ADK;

  bf16x8 qu[NKK], qv[NKK];
  load_q<NKK>(qbase + (long long)(i < Tr ? i : Trc) * ldq, bias_u + h * ADK, qu, i < Tr, lh);
  load_q<NKK>(qbase + (long long)(i < Tr ? i : Trc) * ldq, bias_v + h * ADK, qv, i < Tr, lh);

  f32x16 o[NDT];
#pragma unroll
  for (int t = 0; t < NDT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  float* sg = s_g[wave];
  const uint32_t akey = attn_key(drop, b, h);
  const uint32_t arow = adrop_rcode(akey, (uint32_t)i >> 2), erow = (uint32_t)(i & 3) * (2u * ADROP_G);
  // softmax in the base-2 domain on the RAW scores: p = exp2(raw * c2 - m_raw * c2), c2 = scale * log2(e) > 0 (one fma + v_exp
  // per element instead of mul, sub, mul, v_exp); m_run is the running maximum of the raw scores
  const float c2 = scale * 1.4426950408889634f, tau = 8.f / c2;

  const int nkt = (L + ABK - 1) / ABK;  // key tiles that contain at least one valid key
  // The positional band of step kt is rows c0 + 32*kt .. +159; consecutive steps share four of their five 32-row blocks, so
  // the band lives in a ring of six blocks (block g -> slot g % 6) and a step stages ONE new block (4 KiB instead of 20).
  // K / V tiles are double-buffered.  Everything a step needs is issued one step ahead (dma16: inline-asm LDS-DMA, so that the
  // compiler does not wait for it in front of the step's own LDS reads), behind a single barrier per step.
  const int c0 = T - 1 - (i0_blk + ABQ - 1);
  if (nkt > 0) {
    stage_rows<ADK>(kbase, ldq, 0, Trc, s_k2[0], ABK * NCH);
    stage_v<ADK>(vbase, ldq, 0, Trc, s_v2[0]);
    stage_rows<ADK>(pbase, ldp, c0, P - 1, s_p, ABAND * NCH);
  }
  for (int kt = 0; kt < nkt; ++kt) {
    const int j0 = kt * ABK;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile kt has landed; every wave is done reading tile kt-1
    if (kt + 1 < nkt) {
      stage_rows<ADK>(kbase, ldq, j0 + ABK, Trc, s_k2[(kt + 1) & 1], ABK * NCH);
      stage_v<ADK>(vbase, ldq, j0 + ABK, Trc, s_v2[(kt + 1) & 1]);
      stage_rows<ADK>(pbase, ldp, c0 + 32 * (kt + 5), P - 1, s_p + ((kt + 5) % APRING) * (32 * ADK), 32 * NCH);
    }
    const bf16_t* s_k = s_k2[kt & 1];
    const bf16_t* s_v = s_v2[kt & 1];
    // ring slots of this wave's two band blocks (blocks kt + 3 - wave + gt)
    const int slot0 = (kt + 3 - wave) % APRING, slot1 = (kt + 4 - wave) % APRING;

    // ---- S^T = K . Qu^T  and  G^T = P_band . Qv^T
    f32x16 acc_s, acc_g[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc_s[r] = 0.f; acc_g[0][r] = 0.f; acc_g[1][r] = 0.f; }
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(s_k + a_off<ADK>(q, kk * 2 + lh));
      acc_s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qu[kk], acc_s, 0, 0, 0);
#pragma unroll
      for (int gt = 0; gt < 2; ++gt) {
        const bf16x8 pf = *reinterpret_cast<const bf16x8*>(s_p + a_off<ADK>((gt ? slot1 : slot0) * 32 + q, kk * 2 + lh));
        acc_g[gt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, qv[kk], acc_g[gt], 0, 0, 0);
      }
    }
    // ---- rel_shift through the wave-private LDS tile: G^T[c_local][q] -> sg[q][c_local]; bd[key row] = sg[q][row + 31 - q]
#pragma unroll
    for (int gt = 0; gt < 2; ++gt)
#pragma unroll
      for (int r = 0; r < 16; ++r) sg[q * SG_LD + 32 * gt + (r & 3) + 8 * (r >> 2) + 4 * lh] = acc_g[gt][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s[16];
    float mx = -INFINITY;
    if (j0 + ABK <= L) {
      // every key of the tile is valid (all tiles but the last): no masks.  Lanes of padded queries run on finite numbers
      // and are discarded at the end (inv = 0, lse = 0).
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rho = (r & 3) + 8 * (r >> 2) + 4 * lh;
        s[r] = acc_s[r] + sg[q * SG_LD + rho + 31 - q];
        mx = fmaxf(mx, s[r]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rho = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float bd = sg[q * SG_LD + rho + 31 - q];
        const bool ok = qvalid && (j0 + rho) < L;
        s[r] = ok ? acc_s[r] + bd : -INFINITY;
        mx = fmaxf(mx, s[r]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // Lazy re-scaling: the reference point m_run only moves when some query of the wave finds a score more than 8 (base-2
    // exponent, i.e. a factor 256) above its own -- p <= 256 is harmless in f32 / bf16, l_run carries the same factor and the
    // final normalisation removes it exactly.  After the first tile the accumulators are almost never re-scaled.
    if (__builtin_amdgcn_ballot_w64(mx > m_run + tau) != 0ull) {  // (wave-uniform)
      const float m_new = fmaxf(m_run, mx);
      const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f((m_run - m_new) * c2);  // m_run = -inf -> 0
      l_run *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
      m_run = m_new;
    }
    const float mc = (m_run == -INFINITY) ? 0.f : m_run * c2;
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -mc)); rs += s[r]; }
    rs += __shfl_xor(rs, 32, 64);
    l_run += rs;
    // ---- dropout on the probabilities (the 1/(1-p) scale is folded into the final normalisation)
    if (drop.threshold != 0u) {
      const uint32_t bj0 = (uint32_t)(j0 >> 2);
#pragma unroll
      for (int g = 0; g < 4; ++g) {  // registers 4g..4g+3 = keys 4*bj .. 4*bj+3 of this lane's query, bj = bj0 + 2g + lh
        const uint32_t cc0 = adrop_ccode(bj0 + 2 * g), cc1 = adrop_ccode(bj0 + 2 * g + 1);  // wave-uniform: scalar unit
        const uint32_t seed = adrop_seed(arow ^ (lh ? cc1 : cc0)) + erow;
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
          const uint32_t y = adrop_y(seed + cp * ADROP_G);
          s[4 * g + 2 * cp] = __umul24(y, ADROP_KA) >= drop.threshold ? s[4 * g + 2 * cp] : 0.f;
          s[4 * g + 2 * cp + 1] = __umul24(y, ADROP_KB) >= drop.threshold ? s[4 * g + 2 * cp + 1] : 0.f;
        }
      }
    }
    // ---- O^T += V^T . P^T : B operand = this lane's probabilities (registers 8s..8s+7 <-> key slots of K16 step s)
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const bf16x8 pb = pack8(&s[8 * st]);
#pragma unroll
      for (int dvt = 0; dvt < NDT; ++dvt) {
        const bf16x8 vf = vt_frag<ADK>(s_v, dvt * 32, st, lane);
        o[dvt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb, o[dvt], 0, 0, 0);
      }
    }
  }

  // ---- normalise, transpose O^T -> O through LDS (per wave: [32 queries][64 dv] f32, pitch 66) and store rows
  const float inv = (qvalid && l_run > 0.f) ? (drop.threshold != 0u ? drop.scale : 1.f) / l_run : 0.f;
  __syncthreads();
  if (lse && lh == 0 && i < T) lse[((long long)b * H + h) * T + i] = qvalid ? m_run * scale + __logf(l_run) : 0.f;
#pragma unroll
  for (int half = 0; half < NDT / 2; ++half) {   // the [32][66] tile holds 64 dv columns: wider heads go through it in halves
#pragma unroll
  for (int dvt = 0; dvt < 2; ++dvt)
#pragma unroll
    for (int r = 0; r < 16; ++r) sg[q * SG_LD + dvt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = o[2 * half + dvt][r] * inv;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // 32 rows x 64 dv: lane -> (row = it*8 + lane/8, 8-column chunk = lane%8)
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + (lane >> 3), c8 = (lane & 7) * 8, co = half * 64 + c8;
    const int ii = i0_blk + wave * 32 + row;
    if (ii < Tr) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = sg[row * SG_LD + c8 + j];
      u32x4 t = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
      *reinterpret_cast<u32x4*>(ctx + (row0 + ii) * ldo + h * ADK + co) = t;
      if (ctx_lo) {
        // what the bf16 rounding of O dropped, itself as bf16 (O = hi + lo to ~16 mantissa bits): backward's
        // delta = sum dO * O multiplies a gradient that nearly cancels (dS = P * (dP - delta)); with delta taken from the
        // rounded O alone the q / k / pos_bias_u gradients of near-uniform attention rows lose a digit
        // (profiles/r3_flash_delta.md: 36 % -> 4 % at FastConformer-Large layer 0)
        float lo[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          lo[2 * j] = v[2 * j] - __uint_as_float(t[j] << 16);
          lo[2 * j + 1] = v[2 * j + 1] - __uint_as_float(t[j] & 0xffff0000u);
        }
        u32x4 tl = {pack_bf2(lo[0], lo[1]), pack_bf2(lo[2], lo[3]), pack_bf2(lo[4], lo[5]), pack_bf2(lo[6], lo[7])};
        *reinterpret_cast<u32x4*>(ctx_lo + (row0 + ii) * ldo + h * ADK + co) = tl;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// =================================================================================================
// backward
// =================================================================================================
// delta[b,h,i] = sum_dv dO[i,h,dv] * (O + O_lo)[i,h,dv]   (one wave per row of [M, d]; 8 lanes per head at d_k = 64);
// O_lo (optional) = the rounding residual of O written by the forward kernel
// QB: the same pass also writes qu = q + pos_bias_u and qv = q + pos_bias_v (rows of the fused projection, multi_head_attention.py:
// 288-291), the operands the two backward kernels stage by LDS-DMA -- one launch instead of two in front of every layer's dQ kernel
template <bool QB, int ADK>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ dO, const bf16_t* __restrict__ O,
                                                         const bf16_t* __restrict__ O_lo, float* __restrict__ delta,
                                                         const bf16_t* __restrict__ qkv, long long ldq,
                                                         const float* __restrict__ bias_u, const float* __restrict__ bias_v,
                                                         bf16_t* __restrict__ qu, bf16_t* __restrict__ qv, int B,
                                                         int H, int T, int d, const long long* __restrict__ len,
                                                         const long long* __restrict__ cu) {
  const int lane = threadIdx.x & 63;
  const long long prow = blockIdx.x * 4LL + (threadIdx.x >> 6);   // row of the PADDED [B, T] grid (delta's own layout)
  if (prow >= (long long)B * T) return;
  const int b = (int)(prow / T), i = (int)(prow - (long long)b * T);
  // packed rows: a frame beyond the utterance has no activation row -- its delta is still WRITTEN (zero): the dK/dV kernel stages
  // lse / delta of whole 32-query tiles and multiplies p = 0 by (dP - delta) on the rows beyond the utterance (0 x NaN = NaN)
  const bool rvalid = !cu || i < (int)min((long long)T, len[b]);
  const long long row = cu ? cu[b] + (rvalid ? i : 0) : prow;       // row of the activation matrices
  for (int c0 = 0; c0 < d; c0 += 512) {
    const int c = c0 + lane * 8;
    float acc = 0.f;
    if (QB && c < d && rvalid) {
      float q8[8], u8[8], v8[8], o1[8], o2[8];
      VecIO<bf16_t>::load(qkv + row * ldq + c, q8);
      VecIO<float>::load(bias_u + c, u8); VecIO<float>::load(bias_u + c + 4, &u8[4]);
      VecIO<float>::load(bias_v + c, v8); VecIO<float>::load(bias_v + c + 4, &v8[4]);
#pragma unroll
      for (int j = 0; j < 8; ++j) { o1[j] = q8[j] + u8[j]; o2[j] = q8[j] + v8[j]; }
      VecIO<bf16_t>::store(qu + row * d + c, o1);
      VecIO<bf16_t>::store(qv + row * d + c, o2);
    }
    if (c < d && rvalid) {
      const u32x4 a = *reinterpret_cast<const u32x4*>(dO + row * d + c);
      const u32x4 o = *reinterpret_cast<const u32x4*>(O + row * d + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc += __uint_as_float(a[j] << 16) * __uint_as_float(o[j] << 16);
        acc += __uint_as_float(a[j] & 0xffff0000u) * __uint_as_float(o[j] & 0xffff0000u);
      }
      if (O_lo) {
        const u32x4 l = *reinterpret_cast<const u32x4*>(O_lo + row * d + c);
        float acl = 0.f;  // (the small terms are summed on their own before they meet the large ones)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acl += __uint_as_float(a[j] << 16) * __uint_as_float(l[j] << 16);
          acl += __uint_as_float(a[j] & 0xffff0000u) * __uint_as_float(l[j] & 0xffff0000u);
        }
        acc += acl;
      }
    }
    acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64);
    if (ADK == 128) acc += __shfl_xor(acc, 8, 64);   // ADK / 8 lanes per head
    const int h = c / ADK;
    if ((lane & (ADK / 8 - 1)) == 0 && c < d) delta[((long long)b * H + h) * T + i] = acc;
  }
}

// fragment (8 k-slots) for an MFMA A operand taken TRANSPOSED out of a row-major [row][64] image with the (row>>1)&7
// chunk swizzle: slots 0-3 <- rows ra..ra+3, slots 4-7 <- rows rb..rb+3, all at column col0 + (lane&31)
template <int ADK>
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* img, int ra, int rb, int col0, int lane) {
  const int t = lane & 15, g4 = (lane >> 4) & 1;
  const int col = col0 + g4 * 16 + (t & 3) * 4;
  union { bf16x8 v; s16x4 h[2]; } u;
  const int r0 = ra + (t >> 2), r1 = rb + (t >> 2);
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(img + r0 * ADK + ((((col >> 3) ^ ((r0 >> 1) & 7))) << 3) + (col & 7)));
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(img + r1 * ADK + ((((col >> 3) ^ ((r1 >> 1) & 7))) << 3) + (col & 7)));
  return u.v;
}
// plain row fragments of a [M, d] bf16 matrix as MFMA B operand: lane holds x[row][col0 + kk*16 + (lane>>5)*8 + e]
template <int NKK>
__device__ __forceinline__ void load_rows(const bf16_t* rowp, bf16x8 (&out)[NKK], bool valid, int lh) {
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) {
    union { bf16x8 v; u32x4 w; } u;
    u.w = (u32x4){0u, 0u, 0u, 0u};
    if (valid) u.w = *reinterpret_cast<const u32x4*>(rowp + kk * 16 + lh * 8);
    out[kk] = u.v;
  }
}

// dQu / dQv for one 128-query tile (lane = query, same structure as the forward)
template <int ADK>
__global__ __launch_bounds__(256, ADK == 64 ? 2 : 1) void relpos_flash_bwd_dq_kernel(
    const bf16_t* __restrict__ qu_g, const bf16_t* __restrict__ qv_g, const bf16_t* __restrict__ qkv, long long ldq,
    const bf16_t* __restrict__ pos, long long ldp, const long long* __restrict__ len, const bf16_t* __restrict__ dO,
    const float* __restrict__ lse, const float* __restrict__ delta, bf16_t* __restrict__ dqu_out,
    bf16_t* __restrict__ dqv_out, bf16_t* __restrict__ ds_out, bf16_t* __restrict__ dq_out, long long ld_dq,
    float* __restrict__ cs_partial, int B, int H, int T, int d, float scale, DropCfg drop,
    const long long* __restrict__ cu) {
  drop_resolve(drop);
  constexpr int NKK = ADK / 16, NDT = ADK / 32, NCH = ADK / 8;
  __shared__ __attribute__((aligned(16))) bf16_t s_k2[2][ABK * ADK];       // double-buffered (read until the end of a step)
  __shared__ __attribute__((aligned(16))) bf16_t s_v[ABK * ADK];           // read in the first MFMA block only
  __shared__ __attribute__((aligned(16))) bf16_t s_p[APRING * 32 * ADK];   // positional band: ring of 32-row blocks (see forward)
  __shared__ __attribute__((aligned(16))) float s_g[4][32 * SG_LD];
  __shared__ __attribute__((aligned(16))) bf16_t s_carry[4][32 * 32];  // upper half of the previous step's band, per wave

  const int b = blockIdx.z, h = blockIdx.y;
  const int i0_blk = blockIdx.x * ABQ;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane & 31, lh = lane >> 5;
  const int i = i0_blk + wave * 32 + q;
  const int L = (int)min((long long)T, len[b]);
  const bool qvalid = i < L;
  const int P = 2 * T - 1;
  const long long row0 = cu ? cu[b] : (long long)b * T;   // packed rows: see the forward kernel
  const int Tr = cu ? L : T;
  const int Trc = Tr > 0 ? Tr - 1 : 0;
  if (cu && i0_blk >= L) {
    // packed rows: no query of this tile exists.  Its dS blocks are never read (the linear_pos gradient kernel skips query tiles
    // beyond the utterance); its slab of the bias-gradient column sums must still be zero for the second-stage reduction.
    if (dq_out && cs_partial && threadIdx.x < 2 * ADK) {
      const int uv = threadIdx.x / ADK, c = threadIdx.x % ADK;
      cs_partial[((long long)b * gridDim.x + blockIdx.x) * (2 * d) + uv * d + h * ADK + c] = 0.f;
    }
    return;
  }
  const long long rowi = row0 + (i < Tr ? i : Trc);

  const bf16_t* kbase = qkv + row0 * ldq + h * ADK + (ldq / 3);
  const bf16_t* vbase = kbase + (ldq / 3);
  const bf16_t* pbase = pos + h * ADK;

  bf16x8 qu[NKK], qv[NKK], dof[NKK];
  load_rows<NKK>(qu_g + rowi * d + h * ADK, qu, i < Tr, lh);
  load_rows<NKK>(qv_g + rowi * d + h * ADK, qv, i < Tr, lh);
  load_rows<NKK>(dO + rowi * d + h * ADK, dof, i < Tr, lh);
  float lse_i = qvalid ? lse[((long long)b * H + h) * T + i] : 1e30f;  // padded query: exp(. - 1e30) = 0
  float dlt_i = qvalid ? delta[((long long)b * H + h) * T + i] : 0.f;
  // (the compiler must wait for these register loads HERE, not at their first use inside the loop: see the dK/dV kernel)
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) { asm volatile("" : "+v"(qu[kk])); asm volatile("" : "+v"(qv[kk])); asm volatile("" : "+v"(dof[kk])); }
  asm volatile("" : "+v"(lse_i), "+v"(dlt_i));

  f32x16 dqu[NDT], dqv[NDT];
#pragma unroll
  for (int t = 0; t < NDT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dqu[t][r] = 0.f; dqv[t][r] = 0.f; }
  float* sg = s_g[wave];
  const uint32_t akey = attn_key(drop, b, h);
  const uint32_t arow = adrop_rcode(akey, (uint32_t)i >> 2), erow = (uint32_t)(i & 3) * (2u * ADROP_G);
  // P = exp2(raw * c2 - lse * log2(e)); dS = P * (keep * dP * ks - delta * scale), ks = scale / (1 - p)
  const float c2 = scale * 1.4426950408889634f, lse2 = lse_i * 1.4426950408889634f;
  const float ks = drop.threshold != 0u ? drop.scale * scale : scale, dls = dlt_i * scale;

  const int nkt = (L + ABK - 1) / ABK;
  // linear_pos gradient operand (see the end of the loop): tile `itile` of (h, b) owns nT+1 blocks of 32 x 32 bf16
  const int nT = (T + 31) / 32, itile = blockIdx.x * 4 + wave;
  const bool xw = ds_out != nullptr && itile < nT;  // wave-uniform
  bf16_t* xbase = ds_out + ((((long long)h * B + b) * nT + itile) * (nT + 1)) * 1024;
  u32x4* carry = reinterpret_cast<u32x4*>(s_carry[wave]);
  if (xw) { carry[lane] = (u32x4){0u, 0u, 0u, 0u}; carry[64 + lane] = (u32x4){0u, 0u, 0u, 0u}; }
  // prefetch structure of the forward kernel: K double-buffered and the band's new block one step ahead; V (read in the first
  // MFMA block only) is re-staged into its single buffer behind a second barrier as soon as every wave has read it
  const int c0 = T - 1 - (i0_blk + ABQ - 1);
  if (nkt > 0) {
    stage_rows<ADK>(kbase, ldq, 0, Trc, s_k2[0], ABK * NCH);
    stage_rows<ADK>(vbase, ldq, 0, Trc, s_v, ABK * NCH);
    stage_rows<ADK>(pbase, ldp, c0, P - 1, s_p, ABAND * NCH);
  }
  for (int kt = 0; kt < nkt; ++kt) {
    const int j0 = kt * ABK;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile kt has landed; every wave is done with step kt-1
    if (kt + 1 < nkt) {
      stage_rows<ADK>(kbase, ldq, j0 + ABK, Trc, s_k2[(kt + 1) & 1], ABK * NCH);
      stage_rows<ADK>(pbase, ldp, c0 + 32 * (kt + 5), P - 1, s_p + ((kt + 5) % APRING) * (32 * ADK), 32 * NCH);
    }
    const bf16_t* s_k = s_k2[kt & 1];
    const int slot0 = (kt + 3 - wave) % APRING, slot1 = (kt + 4 - wave) % APRING;  // this wave's two band blocks

    f32x16 acc_s, acc_g[2], acc_dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc_s[r] = 0.f; acc_g[0][r] = 0.f; acc_g[1][r] = 0.f; acc_dp[r] = 0.f; }
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(s_k + a_off<ADK>(q, kk * 2 + lh));
      acc_s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qu[kk], acc_s, 0, 0, 0);
      const bf16x8 vf = *reinterpret_cast<const bf16x8*>(s_v + a_off<ADK>(q, kk * 2 + lh));
      acc_dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[kk], acc_dp, 0, 0, 0);  // dP^T[key][query] = V . dO^T
#pragma unroll
      for (int gt = 0; gt < 2; ++gt) {
        const bf16x8 pf = *reinterpret_cast<const bf16x8*>(s_p + a_off<ADK>((gt ? slot1 : slot0) * 32 + q, kk * 2 + lh));
        acc_g[gt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, qv[kk], acc_g[gt], 0, 0, 0);
      }
    }
    if (kt + 1 < nkt) {  // (block-uniform)
      __syncthreads();   // every wave has its V fragments
      stage_rows<ADK>(vbase, ldq, j0 + ABK, Trc, s_v, ABK * NCH);
    }
#pragma unroll
    for (int gt = 0; gt < 2; ++gt)
#pragma unroll
      for (int r = 0; r < 16; ++r) sg[q * SG_LD + 32 * gt + (r & 3) + 8 * (r >> 2) + 4 * lh] = acc_g[gt][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float ds[16];
    if (j0 + ABK <= L) {  // every key valid; padded queries carry lse_i = +1e30 -> P = 0 without a mask
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rho = (r & 3) + 8 * (r >> 2) + 4 * lh;
        ds[r] = __builtin_amdgcn_exp2f(fmaf(acc_s[r] + sg[q * SG_LD + rho + 31 - q], c2, -lse2));  // P[i, j]
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rho = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float bd = sg[q * SG_LD + rho + 31 - q];
        ds[r] = (j0 + rho) < L ? __builtin_amdgcn_exp2f(fmaf(acc_s[r] + bd, c2, -lse2)) : 0.f;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // dS = P * (dropmask * dP - delta) * scale
    if (drop.threshold != 0u) {
      const uint32_t bj0 = (uint32_t)(j0 >> 2);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint32_t cc0 = adrop_ccode(bj0 + 2 * g), cc1 = adrop_ccode(bj0 + 2 * g + 1);  // wave-uniform: scalar unit
        const uint32_t seed = adrop_seed(arow ^ (lh ? cc1 : cc0)) + erow;
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
          const uint32_t y = adrop_y(seed + cp * ADROP_G);
          const int r0 = 4 * g + 2 * cp;
          const float x0 = __umul24(y, ADROP_KA) >= drop.threshold ? acc_dp[r0] : 0.f;
          const float x1 = __umul24(y, ADROP_KB) >= drop.threshold ? acc_dp[r0 + 1] : 0.f;
          ds[r0] *= fmaf(x0, ks, -dls);
          ds[r0 + 1] *= fmaf(x1, ks, -dls);
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) ds[r] *= fmaf(acc_dp[r], ks, -dls);
    }
    // ---- dQu^T += K^T . dS^T
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const bf16x8 db = pack8(&ds[8 * st]);
#pragma unroll
      for (int dkt = 0; dkt < NDT; ++dkt) {
        const bf16x8 kt_f = tr_frag<ADK>(s_k, 16 * st + 4 * lh, 16 * st + 8 + 4 * lh, dkt * 32, lane);
        dqu[dkt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt_f, db, dqu[dkt], 0, 0, 0);
      }
    }
    // ---- un-skew dS into the band: sg[q][c_local] = dS[q][key = c_local - 31 + q]  (zero outside)
    {
      const float2 z = make_float2(0.f, 0.f);  // rows are 264 B apart: 8-byte aligned only
#pragma unroll
      for (int w2 = 0; w2 < 16; ++w2) *reinterpret_cast<float2*>(sg + q * SG_LD + 32 * lh + 2 * w2) = z;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 16; ++r) sg[q * SG_LD + (r & 3) + 8 * (r >> 2) + 4 * lh + 31 - q] = ds[r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- dQv^T += P_band^T . dG^T   (k-slots: c_local = 16*s4 + 8*lh + e)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      float gv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) gv[e] = sg[q * SG_LD + 16 * s4 + 8 * lh + e];
      const bf16x8 gb = pack8(gv);
#pragma unroll
      for (int dkt = 0; dkt < NDT; ++dkt) {
        const int ra = (s4 < 2 ? slot0 : slot1) * 32 + 16 * (s4 & 1) + 8 * lh;
        const bf16x8 pt_f = tr_frag<ADK>(s_p, ra, ra + 4, dkt * 32, lane);
        dqv[dkt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pt_f, gb, dqv[dkt], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- the band (sg[q][c_local], exactly the operand of the product above) -> HBM for the linear_pos gradient kernel, in
    // the un-shifted (pre-rel_shift) layout: block (it, s) = [32 queries][32 positions c = T-32+32*(s-it) + cl].  The lower
    // half of this step's band (c_local < 32) completes block s = kt together with the upper half of step kt-1 (disjoint
    // triangles: OR of the packed words); the upper half waits in the wave's carry image for step kt+1.  A lane owns the same
    // two 16-byte chunks of the block and of the carry, so the stores are lane-linear (1 KiB per instruction).
    if (xw) {
#pragma unroll
      for (int it2 = 0; it2 < 2; ++it2) {
        const int row = it2 * 16 + (lane >> 2), c8 = (lane & 3) * 8;
        float lo[8], hi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { lo[e] = sg[row * SG_LD + c8 + e]; hi[e] = sg[row * SG_LD + 32 + c8 + e]; }
        const u32x4 cw = carry[it2 * 64 + lane];
        u32x4 t = {pack_bf2(lo[0], lo[1]) | cw.x, pack_bf2(lo[2], lo[3]) | cw.y, pack_bf2(lo[4], lo[5]) | cw.z,
                   pack_bf2(lo[6], lo[7]) | cw.w};
        *reinterpret_cast<u32x4*>(xbase + (long long)kt * 1024 + it2 * 512 + lane * 8) = t;
        carry[it2 * 64 + lane] = (u32x4){pack_bf2(hi[0], hi[1]), pack_bf2(hi[2], hi[3]), pack_bf2(hi[4], hi[5]), pack_bf2(hi[6], hi[7])};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  if (xw) {  // last block of the tile: the upper half of the final step
#pragma unroll
    for (int it2 = 0; it2 < 2; ++it2)
      *reinterpret_cast<u32x4*>(xbase + (long long)nkt * 1024 + it2 * 512 + lane * 8) = carry[it2 * 64 + lane];
  }

  // ---- write dQu, dQv rows (transpose through the wave-private LDS tile).  With dq_out the sum dQ = dQu + dQv goes straight into
  // the q third of the [M, 3d] gradient of the fused projection and the column sums of dQu / dQv over this workgroup's rows
  // (d pos_bias_u / d pos_bias_v, multi_head_attention.py:296-300 backward) into row (b, query block) of `cs_partial`
  // [B * gridDim.x][2 * d] -- a second-stage reduction adds the rows; the separate add + column-sum pass does not exist then.
  __syncthreads();
  constexpr int NH = NDT / 2;   // the [32][66] tile holds 64 columns: wider heads go through it in halves
  float keep[NH][4][8], cs[2][NH][8];
#pragma unroll
  for (int hf = 0; hf < NH; ++hf)
#pragma unroll
    for (int j = 0; j < 8; ++j) { cs[0][hf][j] = 0.f; cs[1][hf][j] = 0.f; }
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    bf16_t* outp = pass == 0 ? dqu_out : dqv_out;
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
#pragma unroll
    for (int dkt = 0; dkt < 2; ++dkt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        sg[q * SG_LD + dkt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = qvalid ? (pass == 0 ? dqu[2 * hf + dkt][r] : dqv[2 * hf + dkt][r]) : 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + (lane >> 3), c8 = (lane & 7) * 8, co = hf * 64 + c8;
      const int ii = i0_blk + wave * 32 + row;
      if (ii < Tr) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = sg[row * SG_LD + c8 + j];
        if (outp) {
          u32x4 t = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
          *reinterpret_cast<u32x4*>(outp + (row0 + ii) * d + h * ADK + co) = t;
        }
        if (dq_out) {
#pragma unroll
          for (int j = 0; j < 8; ++j) cs[pass][hf][j] += v[j];
          if (pass == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) keep[hf][it][j] = v[j];
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += keep[hf][it][j];
            u32x4 t = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
            *reinterpret_cast<u32x4*>(dq_out + (row0 + ii) * ld_dq + h * ADK + co) = t;
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  if (dq_out && cs_partial) {
    // rows of a wave: lanes with equal (lane & 7) hold the same 8 columns -> butterfly over lane bits 3..5, then the four waves
#pragma unroll
    for (int pass = 0; pass < 2; ++pass)
#pragma unroll
      for (int hf = 0; hf < NH; ++hf)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = cs[pass][hf][j];
        v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        cs[pass][hf][j] = v;
      }
    if (lane < 8) {
#pragma unroll
      for (int pass = 0; pass < 2; ++pass)
#pragma unroll
        for (int hf = 0; hf < NH; ++hf)
#pragma unroll
        for (int j = 0; j < 8; ++j) sg[pass * ADK + hf * 64 + lane * 8 + j] = cs[pass][hf][j];
    }
    __syncthreads();
    if (threadIdx.x < 2 * ADK) {
      const int uv = threadIdx.x / ADK, c = threadIdx.x % ADK;
      const float t = (s_g[0][threadIdx.x] + s_g[1][threadIdx.x]) + (s_g[2][threadIdx.x] + s_g[3][threadIdx.x]);
      cs_partial[((long long)b * gridDim.x + blockIdx.x) * (2 * d) + uv * d + h * ADK + c] = t;
    }
  }
}

// dK / dV for one 128-key tile (lane = key): S = Qu K^T, G = Qv P^T with the rel_shift resolved by a lane rotation
// (bd[query rho][key q] = G[rho][31 + q - rho]: a compile-time shift per register), dV^T += dO^T Pd, dK^T += Qu^T dS.
template <int ADK>
__global__ __launch_bounds__(256, ADK == 64 ? 2 : 1) void relpos_flash_bwd_dkv_kernel(
    const bf16_t* __restrict__ qu_g, const bf16_t* __restrict__ qv_g, const bf16_t* __restrict__ qkv, long long ldq,
    const bf16_t* __restrict__ pos, long long ldp, const long long* __restrict__ len, const bf16_t* __restrict__ dO,
    const float* __restrict__ lse, const float* __restrict__ delta, bf16_t* __restrict__ dqkv, long long ldd, int B, int H,
    int T, int Tp, int d, float scale, DropCfg drop, const long long* __restrict__ cu) {
  drop_resolve(drop);
  // staging (double-buffered query tiles + the positional band as a ring of 32-row blocks, see the forward kernel) and the
  // output transposition tile share one buffer: the latter is only used after the loop
  constexpr int NKK = ADK / 16, NDT = ADK / 32, NCH = ADK / 8;
  constexpr int STAGE_BYTES = 2 * 3 * 32 * ADK * 2 + APRING * 32 * ADK * 2 + 2 * 2 * 32 * 4;  // 24 + 24 + 0.5 KiB (ADK = 64)
  constexpr int OUT_BYTES = 4 * 32 * SG_LD * 4;
  __shared__ __attribute__((aligned(16))) char s_raw[STAGE_BYTES > OUT_BYTES ? STAGE_BYTES : OUT_BYTES];
  bf16_t* const s_q3 = reinterpret_cast<bf16_t*>(s_raw);                                 // [2][qu | qv | dO][32 * ADK]
  bf16_t* const s_p = reinterpret_cast<bf16_t*>(s_raw + 2 * 3 * 32 * ADK * 2);          // [APRING * 32][ADK]
  float* const s_ld = reinterpret_cast<float*>(s_raw + 2 * 3 * 32 * ADK * 2 + APRING * 32 * ADK * 2);  // [2][lse | delta][32]

  const int b = blockIdx.z, h = blockIdx.y;
  const int j0_blk = blockIdx.x * ABQ;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane & 31, lh = lane >> 5;
  const int j = j0_blk + wave * 32 + q;  // this lane's key
  const int L = (int)min((long long)T, len[b]);
  const bool kvalid = j < L;
  const int P = 2 * T - 1;
  const long long row0 = cu ? cu[b] : (long long)b * T;   // packed rows: see the forward kernel
  const int Tr = cu ? L : T;
  const int Trc = Tr > 0 ? Tr - 1 : 0;
  if (cu && j0_blk >= L) return;   // packed rows: no key of this tile exists, nothing to write
  const long long rowj = row0 + (j < Tr ? j : Trc);
  const bf16_t* kbase = qkv + (ldq / 3) + h * ADK;
  const bf16_t* pbase = pos + h * ADK;
  const uint32_t akey = attn_key(drop, b, h);
  const uint32_t acol = adrop_ccode((uint32_t)j >> 2), ecol = (uint32_t)((j & 3) >> 1) * ADROP_G;
  const uint32_t klane = (j & 1) ? ADROP_KB : ADROP_KA;  // this key column's multiplier (see the dropout definition)
  const float dscale = drop.threshold != 0u ? drop.scale : 1.f;  // folded, with `scale`, into the accumulators at the end

  bf16x8 kf[NKK], vf[NKK];
  load_rows<NKK>(kbase + rowj * ldq, kf, j < Tr, lh);
  load_rows<NKK>(kbase + (ldq / 3) + rowj * ldq, vf, j < Tr, lh);

  // (the compiler must wait for these register loads HERE: inside the loop its `s_waitcnt vmcnt(0)` at their first use would
  //  also wait, every step, for the prefetch it knows nothing about)
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) { asm volatile("" : "+v"(kf[kk])); asm volatile("" : "+v"(vf[kk])); }
  f32x16 dk_acc[NDT], dv_acc[NDT];
#pragma unroll
  for (int t = 0; t < NDT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk_acc[t][r] = 0.f; dv_acc[t][r] = 0.f; }

  const int nqt = (L + 31) / 32;
  const bf16_t* qub = qu_g + row0 * d + h * ADK;
  const bf16_t* qvb = qv_g + row0 * d + h * ADK;
  const bf16_t* dob = dO + row0 * d + h * ADK;
  // band rows of step qt: cb0 - 32*qt .. +159 (blocks -qt .. -qt+4 of 32 rows, block g in ring slot (g mod 6)); the step after
  // needs ONE new block at the low end.  Query tiles (Qu, Qv, dO) and their lse / delta (wave 0: lanes 0-31 | 32-63, 4 bytes
  // each) are double-buffered.
  const int cb0 = T - 1 + j0_blk - 31;
  auto stage_q = [&](int i0, int buf) {
    bf16_t* dst = s_q3 + buf * (3 * 32 * ADK);
    stage_rows<ADK>(qub, d, i0, Trc, dst, 32 * NCH);
    stage_rows<ADK>(qvb, d, i0, Trc, dst + 32 * ADK, 32 * NCH);
    stage_rows<ADK>(dob, d, i0, Trc, dst + 2 * 32 * ADK, 32 * NCH);
    if (wave == 0) {
      int ii = i0 + (lane & 31);
      ii = ii < T ? ii : T - 1;  // (rows >= len are never used: p = 0 there)
      dma4((lane < 32 ? lse : delta) + ((long long)b * H + h) * T + ii, s_ld + buf * 64);
    }
  };
  if (nqt > 0) {
    stage_q(0, 0);
    stage_rows<ADK>(pbase, ldp, cb0, P - 1, s_p, ABAND * NCH);
  }
  for (int qt = 0; qt < nqt; ++qt) {
    const int i0 = qt * 32;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile qt has landed; every wave is done with step qt-1
    if (qt + 1 < nqt) {
      stage_q(i0 + 32, (qt + 1) & 1);
      stage_rows<ADK>(pbase, ldp, cb0 - 32 * (qt + 1), P - 1, s_p + ((6 * 1024 - (qt + 1)) % APRING) * (32 * ADK), 32 * NCH);
    }
    const bf16_t* s_qu = s_q3 + (qt & 1) * (3 * 32 * ADK);
    const bf16_t* s_qv = s_qu + 32 * ADK;
    const bf16_t* s_do = s_qu + 2 * 32 * ADK;
    const float* s_lse = s_ld + (qt & 1) * 64;
    const float* s_dlt = s_lse + 32;
    const int slot0 = (6 * 1024 - qt + wave) % APRING, slot1 = (6 * 1024 - qt + wave + 1) % APRING;  // band blocks wave, wave + 1

    f32x16 acc_s, acc_g[2], acc_dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc_s[r] = 0.f; acc_g[0][r] = 0.f; acc_g[1][r] = 0.f; acc_dp[r] = 0.f; }
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      const bf16x8 quf = *reinterpret_cast<const bf16x8*>(s_qu + a_off<ADK>(q, kk * 2 + lh));
      acc_s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(quf, kf[kk], acc_s, 0, 0, 0);     // S[query][key]
      const bf16x8 dof = *reinterpret_cast<const bf16x8*>(s_do + a_off<ADK>(q, kk * 2 + lh));
      acc_dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof, vf[kk], acc_dp, 0, 0, 0);   // dP[query][key]
      const bf16x8 qvf = *reinterpret_cast<const bf16x8*>(s_qv + a_off<ADK>(q, kk * 2 + lh));
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const bf16x8 pf = *reinterpret_cast<const bf16x8*>(s_p + a_off<ADK>((ct ? slot1 : slot0) * 32 + q, kk * 2 + lh));
        acc_g[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qvf, pf, acc_g[ct], 0, 0, 0);  // G[query][c_local]
      }
    }
    float pd[16], ds[16];
    uint32_t w4[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};  // keep iff w >= threshold (threshold 0: dropout off)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if ((r & 3) == 0 && drop.threshold != 0u) {  // registers 4g..4g+3 = queries 4*bi .. 4*bi+3 against this lane's key
        const uint32_t rr0 = adrop_rcode(akey, (uint32_t)(i0 >> 2) + 2 * (r >> 2));      // bi = i0/4 + 2g + lh: both candidates
        const uint32_t rr1 = adrop_rcode(akey, (uint32_t)(i0 >> 2) + 2 * (r >> 2) + 1);  // are wave-uniform (scalar unit)
        const uint32_t seed = adrop_seed((lh ? rr1 : rr0) ^ acol) + ecol;
#pragma unroll
        for (int c = 0; c < 4; ++c) w4[c] = __umul24(adrop_y(seed + c * (2u * ADROP_G)), klane);
      }
      const int rho = (r & 3) + 8 * (r >> 2) + 4 * lh;  // query row of this register
      const int sl = 31 + q - rho;                      // column of G that holds c(i, j)
      const float g0 = __shfl(acc_g[0][r], lh * 32 + (sl & 31), 64);
      const float g1 = __shfl(acc_g[1][r], lh * 32 + (sl & 31), 64);
      const float bd = sl < 32 ? g0 : g1;
      const int ii = i0 + rho;
      const bool ok = kvalid && ii < L;
      const float p = ok ? __expf((acc_s[r] + bd) * scale - s_lse[rho]) : 0.f;
      const bool keep = w4[r & 3] >= drop.threshold;
      pd[r] = keep ? p : 0.f;
      ds[r] = p * fmaf(keep ? acc_dp[r] : 0.f, dscale, -s_dlt[rho]);  // (x scale: applied to dK at the end)
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const bf16x8 pb = pack8(&pd[8 * st]);
      const bf16x8 db = pack8(&ds[8 * st]);
      const int ra = 16 * st + 4 * lh, rb = 16 * st + 8 + 4 * lh;
#pragma unroll
      for (int t2 = 0; t2 < NDT; ++t2) {
        const bf16x8 dot_f = tr_frag<ADK>(s_do, ra, rb, t2 * 32, lane);   // dO^T[dv][query slots]
        dv_acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dot_f, pb, dv_acc[t2], 0, 0, 0);
        const bf16x8 qut_f = tr_frag<ADK>(s_qu, ra, rb, t2 * 32, lane);   // Qu^T[dk][query slots]
        dk_acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qut_f, db, dk_acc[t2], 0, 0, 0);
      }
    }
  }
  __syncthreads();  // the transposition tile below overlays the staging buffers

  // ---- write dK, dV rows of this wave's 32 keys
  float* st_ = reinterpret_cast<float*>(s_raw) + wave * (32 * SG_LD);
#pragma unroll
  for (int pass = 0; pass < 2 * (NDT / 2); ++pass) {   // (dK | dV) x 64-column halves through the [32][66] tile
    const int kind = pass / (NDT / 2), hf = pass % (NDT / 2);
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        st_[q * SG_LD + t2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] =
            kind == 0 ? dk_acc[2 * hf + t2][r] * scale : dv_acc[2 * hf + t2][r] * dscale;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + (lane >> 3), c8 = (lane & 7) * 8;
      const int jj = j0_blk + wave * 32 + row;
      if (jj < Tr) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = st_[row * SG_LD + c8 + e];
        u32x4 t = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
        *reinterpret_cast<u32x4*>(dqkv + (row0 + jj) * ldd + (kind + 1) * (ldd / 3) + h * ADK + hf * 64 + c8) = t;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// 32 x 32 bf16 image [row = k][col = m] (pitch 32, no swizzle: the 4 rows x 32 columns a half-wave reads are 256 contiguous
// bytes) -> MFMA A fragment A[m = lane&31][k-slots 0-3 <- rows ra..ra+3, 4-7 <- rows rb..rb+3] by the hardware transpose read
__device__ __forceinline__ bf16x8 tr_frag32(const bf16_t* img, int ra, int rb, int lane) {
  const int t = lane & 15, g4 = (lane >> 4) & 1;
  const int col = g4 * 16 + (t & 3) * 4;
  union { bf16x8 v; s16x4 h[2]; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(img + (ra + (t >> 2)) * 32 + col));
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(img + (rb + (t >> 2)) * 32 + col));
  return u.v;
}

// d linear_pos output: dp[c, h, dk] += sum_{b,i} dS[b,h,i, j = c-(T-1)+i] * Qv[b,i,h,dk].  The dQ kernel leaves dS in the
// UN-SHIFTED layout of the reference's matrix_bd before rel_shift (multi_head_attention.py:259-270), cut into blocks:
// X[h][b][it][s] = [32 queries of tile it][32 positions c = T-32+32*(s-it) + cl], s = 0..nT.  That makes the gradient a plain
// product with both operands staged as they lie in memory:
//   D[c][dk] += sum_q X[q][c] * Qv[q][dk]     (A = X^T and B = Qv both through ds_read_b64_tr_b16, no gather, no masks)
// One workgroup owns 64 consecutive positions (the block pair dg = s-it in {2x-(nT-1), +1}) of one head and walks every
// (utterance of its chunk, query tile) that reaches them: 4 KiB of X (the two blocks are adjacent) + 4 KiB of Qv per item by
// LDS-DMA one item ahead, 8 MFMAs; the four waves split the items and combine at the end.
template <int ADK>
__global__ __launch_bounds__(256, ADK == 64 ? 2 : 1) void relpos_flash_bwd_dpos_kernel(
    const bf16_t* __restrict__ qv_g, const bf16_t* __restrict__ x_g, const long long* __restrict__ len,
    float* __restrict__ dpos, long long ldd, float* __restrict__ partial, int B, int H, int T, int d, int bchunk,
    const long long* __restrict__ cu) {
  constexpr int NDT = ADK / 32, NCH = ADK / 8;
  __shared__ __attribute__((aligned(16))) bf16_t s_x[4][2][2 * 1024];   // per wave, double-buffered
  __shared__ __attribute__((aligned(16))) bf16_t s_qv[4][2][32 * ADK];

  const int nT = (T + 31) / 32;
  const int dg = 2 * (int)blockIdx.x - (nT - 1);  // s - it of the first block; the second is dg + 1
  const int h = blockIdx.y;
  const int b_begin = blockIdx.z * bchunk, b_end = min(B, b_begin + bchunk);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane & 31, lh = lane >> 5;
  const int P = 2 * T - 1;
  const int cmin = T - 32 + 32 * dg;
  // query tiles with at least one of the slots it+dg, it+dg+1 inside [0, nT]
  const int it_lo = max(0, -dg - 1), it_hi = min(nT, nT - dg + 1);

  f32x16 dp_acc[2][NDT];
#pragma unroll
  for (int t = 0; t < NDT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dp_acc[0][t][r] = 0.f; dp_acc[1][t][r] = 0.f; }

  const int npairs = it_hi - it_lo;
  const int nitems = (b_end - b_begin) * npairs;
  // 8 LDS-DMA instructions per item; a slot outside [0, nT] is clamped for the load (always inside the tile's own blocks) and
  // skipped in the product; batch / head offsets are folded into 64-bit bases once per item
  auto issue = [&](int b, int it, int buf) {
    const int s0 = it + dg;
    const int sl0 = min(max(s0, 0), nT), sl1 = min(max(s0 + 1, 0), nT);
    const bf16_t* base = x_g + ((((long long)h * B + b) * nT + it) * (nT + 1)) * 1024;
    bf16_t* sx = s_x[wave][buf];
    bf16_t* sqv = s_qv[wave][buf];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      dma16((base + sl0 * 1024 + k2 * 512 + lane * 8), (sx + k2 * 512));
      dma16((base + sl1 * 1024 + k2 * 512 + lane * 8), (sx + 1024 + k2 * 512));
    }
    // packed rows (cu): utterance b owns rows cu[b] .. cu[b] + L - 1 of qv; rows beyond meet zero blocks of X (any finite row does)
    const int Lb = cu ? (int)min((long long)T, len[b]) : T;
    const bf16_t* qb = qv_g + (cu ? (Lb > 0 ? cu[b] : 0LL) : (long long)b * T) * d + h * ADK;
    const int grmax = Lb > 0 ? Lb - 1 : 0;
#pragma unroll
    for (int k4 = 0; k4 < 32 * NCH / 64; ++k4) {
      const int cq = k4 * 64 + lane;
      const int r = cq / NCH, ck = cq % NCH;
      int gr = 32 * it + r;
      gr = gr > grmax ? grmax : gr;
      dma16((qb + __mul24(gr, d) + ((ck ^ ((r >> 1) & 7)) << 3)), (sqv + k4 * 512));
    }
  };
  auto advance = [&](int& b, int& it, int steps) {  // steps <= 4 < npairs is not guaranteed: loop
    it += steps;
    while (it >= it_hi) { it -= npairs; ++b; }
  };
  // per-lane LDS byte addresses of the transpose reads (tr_frag32 / tr_frag geometry), buffer 0
  uint32_t xaddr, qaddr[2][NDT][2];
  {
    const int t = lane & 15, g4 = (lane >> 4) & 1;
    xaddr = lds_addr(&s_x[wave][0][0]) + (uint32_t)(((8 * lh + (t >> 2)) * 32 + g4 * 16 + (t & 3) * 4) * 2);
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int dkt = 0; dkt < NDT; ++dkt)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int r = 16 * st + 8 * lh + 4 * hf + (t >> 2), col = dkt * 32 + g4 * 16 + (t & 3) * 4;
          qaddr[st][dkt][hf] = lds_addr(&s_qv[wave][0][0]) + (uint32_t)((r * ADK + (((col >> 3) ^ ((r >> 1) & 7)) << 3) + (col & 7)) * 2);
        }
  }
  int cb = b_begin, cit = it_lo;      // item being computed by this wave
  advance(cb, cit, wave);
  int nb = cb, nit = cit;             // item being prefetched
  if (wave < nitems) issue(cb, cit, 0);
  int buf = 0;
  for (int item = wave; item < nitems; item += 4, buf ^= 1, advance(cb, cit, 4)) {
    if (item + 4 < nitems) {
      nb = cb; nit = cit; advance(nb, nit, 4);
      issue(nb, nit, buf ^ 1);
      if constexpr (ADK == 64) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // (the next item's 4 + 4 loads stay in flight)
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                       // (4 + 8)
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int b = cb;
    const int it = cit, s0 = it + dg;
    const int L = (int)min((long long)T, len[b]);
    const int nkt = (L + 31) / 32;  // the dQ kernel wrote slots 0..nkt of every tile
    const bool v0 = s0 >= 0 && s0 <= nkt, v1 = s0 + 1 >= 0 && s0 + 1 <= nkt;
    if (32 * it >= L || !(v0 || v1)) continue;  // wave-uniform: tiles without valid queries are all zero, other slots unwritten
    // fragment reads by inline asm: behind an LDS-DMA the compiler puts `s_waitcnt vmcnt(0)` in front of every LDS read it
    // emits itself, which would also wait for the NEXT item's eight loads (the counted vmcnt(8) above is the real condition)
    const uint32_t m0 = v0 ? 0xffffffffu : 0u, m1 = v1 ? 0xffffffffu : 0u;  // a slot outside 0..nkt was never written
    const uint32_t xa = xaddr + buf * 4096u, qoff = buf * (uint32_t)(32 * ADK * 2);
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      union { bf16x8 v; s16x4 h[2]; u32x4 w; } af[2], bq[NDT];
      asm volatile(
          "ds_read_b64_tr_b16 %0, %8\n\t"
          "ds_read_b64_tr_b16 %1, %8 offset:256\n\t"
          "ds_read_b64_tr_b16 %2, %8 offset:2048\n\t"
          "ds_read_b64_tr_b16 %3, %8 offset:2304\n\t"
          "ds_read_b64_tr_b16 %4, %9\n\t"
          "ds_read_b64_tr_b16 %5, %10\n\t"
          "ds_read_b64_tr_b16 %6, %11\n\t"
          "ds_read_b64_tr_b16 %7, %12\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(af[0].h[0]), "=&v"(af[0].h[1]), "=&v"(af[1].h[0]), "=&v"(af[1].h[1]), "=&v"(bq[0].h[0]), "=&v"(bq[0].h[1]),
            "=&v"(bq[1].h[0]), "=&v"(bq[1].h[1])
          : "v"(xa + st * 1024u), "v"(qaddr[st][0][0] + qoff), "v"(qaddr[st][0][1] + qoff), "v"(qaddr[st][1][0] + qoff),
            "v"(qaddr[st][1][1] + qoff)
          : "memory");
      if constexpr (NDT == 4) {
        asm volatile(
            "ds_read_b64_tr_b16 %0, %4\n\t"
            "ds_read_b64_tr_b16 %1, %5\n\t"
            "ds_read_b64_tr_b16 %2, %6\n\t"
            "ds_read_b64_tr_b16 %3, %7\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(bq[2].h[0]), "=&v"(bq[2].h[1]), "=&v"(bq[3].h[0]), "=&v"(bq[3].h[1])
            : "v"(qaddr[st][2][0] + qoff), "v"(qaddr[st][2][1] + qoff), "v"(qaddr[st][3][0] + qoff), "v"(qaddr[st][3][1] + qoff)
            : "memory");
      }
      af[0].w &= m0;
      af[1].w &= m1;
#pragma unroll
      for (int dkt = 0; dkt < NDT; ++dkt) {
        dp_acc[0][dkt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0].v, bq[dkt].v, dp_acc[0][dkt], 0, 0, 0);
        dp_acc[1][dkt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1].v, bq[dkt].v, dp_acc[1][dkt], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  // ---- combine the 4 waves with plain LDS traffic (ds_add_f32 atomics measured ~10 us per workgroup here): waves 2,3 park
  // their accumulators in two slabs, waves 0,1 add them, wave 1 parks, wave 0 adds and owns the result.
  __syncthreads();
  float* slab = reinterpret_cast<float*>(&s_qv[0][0][0]);  // 512 x ADK bytes = 2 slabs of 64 x ADK floats
  constexpr int SLAB = 64 * ADK;
  if (wave >= 2) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int dkt = 0; dkt < NDT; ++dkt)
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[(wave - 2) * SLAB + ((ct * NDT + dkt) * 16 + r) * 64 + lane] = dp_acc[ct][dkt][r];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int dkt = 0; dkt < NDT; ++dkt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dp_acc[ct][dkt][r] += slab[wave * SLAB + ((ct * NDT + dkt) * 16 + r) * 64 + lane];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int dkt = 0; dkt < NDT; ++dkt)
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[((ct * NDT + dkt) * 16 + r) * 64 + lane] = dp_acc[ct][dkt][r];
  }
  __syncthreads();
  if (wave != 0) return;
  float* out_slab = partial ? partial + (((long long)blockIdx.z * gridDim.x + blockIdx.x) * H + h) * SLAB : nullptr;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int dkt = 0; dkt < NDT; ++dkt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = dp_acc[ct][dkt][r] + slab[((ct * NDT + dkt) * 16 + r) * 64 + lane];
        const int cl = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, dk = dkt * 32 + q;
        if (out_slab) out_slab[cl * ADK + dk] = v;  // deterministic two-stage reduction (dpos_reduce_kernel)
        else {
          const int c = cmin + cl;
          if (c >= 0 && c < P) atomicAdd(dpos + (long long)c * ldd + h * ADK + dk, v);
        }
      }
}

// stage 2: the 64 positions of block pair x (disjoint between pairs), summed over the utterance chunks z
template <int ADK>
__global__ __launch_bounds__(256) void dpos_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dpos, long long ldd,
                                                          bf16_t* __restrict__ dpos_cast, int H, int T, int nz) {
  const int nT = (T + 31) / 32;
  const int x = blockIdx.x, h = blockIdx.y;
  const int P = 2 * T - 1;
  const int e = blockIdx.z * 256 + threadIdx.x;  // ADK / 4 blocks of 256 threads cover the 64 x ADK pair
  const int cl = e / ADK, dk = e % ADK;
  const int c = T - 32 + 32 * (2 * x - (nT - 1)) + cl;
  if (c < 0 || c >= P) return;
  float acc = 0.f;
  for (int z = 0; z < nz; ++z) acc += partial[(((long long)z * nT + x) * H + h) * (64 * ADK) + e];
  const float v = dpos[(long long)c * ldd + h * ADK + dk] + acc;
  dpos[(long long)c * ldd + h * ADK + dk] = v;
  // every (c, h, dk) of [0, 2T-1) x H x 64 is owned by exactly one thread of this launch, so the GEMM-operand copy of the
  // gradient (the linear_pos weight gradient's input) can be written here instead of by a cast pass of its own
  if (dpos_cast) dpos_cast[(long long)c * ldd + h * ADK + dk] = f2bf(v);
}

extern "C" int mi355x_relpos_flash_fwd(const void* qkv, long long ldq, const void* pos, long long ldp, const void* bias_u,
                                       const void* bias_v, const void* len, void* ctx, void* ctx_lo, long long ldo, void* lse,
                                       int B, int H, int T, int dk, int Tp, float scale, unsigned drop_key,
                                       unsigned drop_threshold, float drop_scale, const void* row_offsets, void* stream) {
  mi_clear_errors();
  if (!qkv || !pos || !bias_u || !bias_v || !len || !ctx || B <= 0 || H <= 0 || T <= 0) return MI_ERR_ARG;
  if ((dk != 64 && dk != 128) || (ldq % 24) || (ldp & 7) || (ldo & 7) || ((uintptr_t)qkv & 15) || ((uintptr_t)pos & 15) ||
      ((uintptr_t)ctx & 15) || ((uintptr_t)ctx_lo & 15))
    return MI_ERR_ARG;
  DropCfg dc = mi_drop(drop_key, drop_threshold, drop_scale);
  dim3 grid((T + ABQ - 1) / ABQ, H, B);
#define FWD_LAUNCH(DK)                                                                                                              \
  MI_LAUNCH(relpos_flash_fwd_kernel<DK>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, ldq, (const bf16_t*)pos, ldp, \
            (const float*)bias_u, (const float*)bias_v, (const long long*)len, (bf16_t*)ctx, (bf16_t*)ctx_lo, ldo, (float*)lse, B, \
            H, T, Tp, scale, dc, (const long long*)row_offsets)
  if (dk == 64) { FWD_LAUNCH(64); } else { FWD_LAUNCH(128); }
#undef FWD_LAUNCH
  return mi_check_launch();
}

extern "C" int mi355x_attn_delta(const void* dO, const void* O, const void* O_lo, void* delta, int B, int H, int T, int d,
                                 const void* len, const void* row_offsets, void* stream) {
  mi_clear_errors();
  if (!dO || !O || !delta || B <= 0 || H <= 0 || T <= 0 || (d != H * 64 && d != H * 128) || (row_offsets && !len)) return MI_ERR_ARG;
  const long long rows = (long long)B * T;
#define DELTA_LAUNCH(DK)                                                                                                  \
  MI_LAUNCH((attn_delta_kernel<false, DK>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,        \
            (const bf16_t*)dO, (const bf16_t*)O, (const bf16_t*)O_lo, (float*)delta, (const bf16_t*)nullptr, 0LL,         \
            (const float*)nullptr, (const float*)nullptr, (bf16_t*)nullptr, (bf16_t*)nullptr, B, H, T, d,                 \
            (const long long*)len, (const long long*)row_offsets)
  if (d == H * 64) { DELTA_LAUNCH(64); } else { DELTA_LAUNCH(128); }
#undef DELTA_LAUNCH
  return mi_check_launch();
}
extern "C" int mi355x_attn_bwd_prep(const void* dO, const void* O, const void* O_lo, void* delta, const void* qkv, long long ldq,
                                    const void* bias_u, const void* bias_v, void* qu, void* qv, int B, int H, int T, int d,
                                    const void* len, const void* row_offsets, void* stream) {
  mi_clear_errors();
  if (!dO || !O || !delta || !qkv || !bias_u || !bias_v || !qu || !qv || B <= 0 || H <= 0 || T <= 0 ||
      (d != H * 64 && d != H * 128) || (ldq & 7) || (row_offsets && !len))
    return MI_ERR_ARG;
  if (((uintptr_t)qkv | (uintptr_t)qu | (uintptr_t)qv | (uintptr_t)bias_u | (uintptr_t)bias_v) & 15) return MI_ERR_ARG;
  const long long rows = (long long)B * T;
#define PREP_LAUNCH(DK)                                                                                                   \
  MI_LAUNCH((attn_delta_kernel<true, DK>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,         \
            (const bf16_t*)dO, (const bf16_t*)O, (const bf16_t*)O_lo, (float*)delta, (const bf16_t*)qkv, ldq,             \
            (const float*)bias_u, (const float*)bias_v, (bf16_t*)qu, (bf16_t*)qv, B, H, T, d, (const long long*)len,      \
            (const long long*)row_offsets)
  if (d == H * 64) { PREP_LAUNCH(64); } else { PREP_LAUNCH(128); }
#undef PREP_LAUNCH
  return mi_check_launch();
}

extern "C" int mi355x_relpos_flash_bwd_dq(const void* qu, const void* qv, const void* qkv, long long ldq, const void* pos,
                                          long long ldp, const void* len, const void* dO, const void* lse, const void* delta,
                                          void* dqu, void* dqv, void* ds_out, void* dq_out, long long ld_dq, void* bias_grads,
                                          void* cs_scratch, long long cs_scratch_elems, int B, int H, int T, int dk,
                                          long long ds_elems, float scale, unsigned drop_key, unsigned drop_threshold,
                                          float drop_scale, const void* row_offsets, void* stream) {
  mi_clear_errors();
  if (!qu || !qv || !qkv || !pos || !len || !dO || !lse || !delta || B <= 0 || H <= 0 || T <= 0) return MI_ERR_ARG;
  if ((dk != 64 && dk != 128) || (ldq % 24) || (ldp & 7)) return MI_ERR_ARG;
  // either the two gradients separately (dqu, dqv) or their sum (dq_out, row pitch ld_dq) with the bias gradients
  // (bias_grads f32 [2 * H * 64] += column sums of dQu | dQv, via cs_scratch: f32 [B * ceil(T / 128) * 2 * H * 64])
  if (!dq_out && (!dqu || !dqv)) return MI_ERR_ARG;
  const long long cs_need = (long long)B * ((T + ABQ - 1) / ABQ) * 2 * H * dk;
  if (dq_out && ((ld_dq & 7) || ((uintptr_t)dq_out & 15) || (bias_grads && (!cs_scratch || cs_scratch_elems < cs_need))))
    return MI_ERR_ARG;
  if (ds_out && (((uintptr_t)ds_out & 15) || ds_elems < mi355x_relpos_ds_elems(B, H, T))) return MI_ERR_ARG;
  DropCfg dc = mi_drop(drop_key, drop_threshold, drop_scale);
  dim3 grid((T + ABQ - 1) / ABQ, H, B);
#define DQ_LAUNCH(DK)                                                                                                              \
  MI_LAUNCH(relpos_flash_bwd_dq_kernel<DK>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qu, (const bf16_t*)qv,           \
            (const bf16_t*)qkv, ldq, (const bf16_t*)pos, ldp, (const long long*)len, (const bf16_t*)dO, (const float*)lse,           \
            (const float*)delta, (bf16_t*)dqu, (bf16_t*)dqv, (bf16_t*)ds_out, (bf16_t*)dq_out, ld_dq,                               \
            bias_grads ? (float*)cs_scratch : nullptr, B, H, T, H * DK, scale, dc, (const long long*)row_offsets)
  if (dk == 64) { DQ_LAUNCH(64); } else { DQ_LAUNCH(128); }
#undef DQ_LAUNCH
  if (dq_out && bias_grads)
    MI_LAUNCH((partials_reduce_kernel<float>), dim3((2 * H * dk + 255) / 256, 8), dim3(256), 0, (hipStream_t)stream,
              (const float*)cs_scratch, (int)(B * ((T + ABQ - 1) / ABQ)), 2 * H * dk, (float*)bias_grads);
  return mi_check_launch();
}

extern "C" int mi355x_relpos_flash_bwd_dkv(const void* qu, const void* qv, const void* qkv, long long ldq, const void* pos,
                                           long long ldp, const void* len, const void* dO, const void* lse, const void* delta,
                                           void* dqkv, long long ldd, int B, int H, int T, int dk, int Tp, float scale,
                                           unsigned drop_key, unsigned drop_threshold, float drop_scale, const void* row_offsets,
                                           void* stream) {
  mi_clear_errors();
  if (!qu || !qv || !qkv || !pos || !len || !dO || !lse || !delta || !dqkv || B <= 0 || H <= 0 || T <= 0) return MI_ERR_ARG;
  if ((dk != 64 && dk != 128) || (ldq % 24) || (ldp & 7) || (ldd % 24)) return MI_ERR_ARG;
  DropCfg dc = mi_drop(drop_key, drop_threshold, drop_scale);
  dim3 grid((T + ABQ - 1) / ABQ, H, B);
#define DKV_LAUNCH(DK)                                                                                                             \
  MI_LAUNCH(relpos_flash_bwd_dkv_kernel<DK>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qu, (const bf16_t*)qv,         \
            (const bf16_t*)qkv, ldq, (const bf16_t*)pos, ldp, (const long long*)len, (const bf16_t*)dO, (const float*)lse,          \
            (const float*)delta, (bf16_t*)dqkv, ldd, B, H, T, Tp, H * DK, scale, dc, (const long long*)row_offsets)
  if (dk == 64) { DKV_LAUNCH(64); } else { DKV_LAUNCH(128); }
#undef DKV_LAUNCH
  return mi_check_launch();
}

extern "C" long long mi355x_relpos_ds_elems(int B, int H, int T) {
  const long long nT = (T + 31) / 32;
  return (long long)H * B * nT * (nT + 1) * 1024;
}

extern "C" long long mi355x_relpos_dpos_partial_elems(int B, int H, int T) {
  const long long nT = (T + 31) / 32;
  const int bchunk = B >= 8 ? 4 : 1;
  return (long long)((B + bchunk - 1) / bchunk) * nT * H * (64 * 128);   // 64 positions x the widest head (d_k' = 128)
}

extern "C" int mi355x_relpos_flash_bwd_dpos(const void* qv, const void* ds, const void* len, void* dpos, long long ldd,
                                            void* dpos_cast, void* partial, long long partial_elems, int B, int H, int T, int dk,
                                            long long ds_elems, const void* row_offsets, void* stream) {
  mi_clear_errors();
  if (!qv || !ds || !len || !dpos || B <= 0 || H <= 0 || T <= 0 || (dpos_cast && !partial)) return MI_ERR_ARG;
  if ((dk != 64 && dk != 128) || ((uintptr_t)ds & 15) || ds_elems < mi355x_relpos_ds_elems(B, H, T)) return MI_ERR_ARG;
  const int nT = (T + 31) / 32;
  const int bchunk = B >= 8 ? 4 : 1;
  const int nz = (B + bchunk - 1) / bchunk;
  if (partial && partial_elems < mi355x_relpos_dpos_partial_elems(B, H, T)) return MI_ERR_ARG;
  dim3 grid(nT, H, nz);
#define DPOS_LAUNCH(DK)                                                                                                        \
  MI_LAUNCH(relpos_flash_bwd_dpos_kernel<DK>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qv, (const bf16_t*)ds,   \
            (const long long*)len, (float*)dpos, ldd, (float*)partial, B, H, T, H * DK, bchunk, (const long long*)row_offsets); \
  if (partial)                                                                                                                 \
    MI_LAUNCH(dpos_reduce_kernel<DK>, dim3(nT, H, DK / 4), dim3(256), 0, (hipStream_t)stream, (const float*)partial,           \
              (float*)dpos, ldd, (bf16_t*)dpos_cast, H, T, nz)
  if (dk == 64) { DPOS_LAUNCH(64); } else { DPOS_LAUNCH(128); }
#undef DPOS_LAUNCH
  return mi_check_launch();
}
