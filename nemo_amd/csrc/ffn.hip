// Fused macaron feed-forward block of the Conformer layer for d_model = 512 (Conformer-CTC-Large, FastConformer-Large) on
// MI355X (gfx950): ONE launch per direction instead of two GEMM launches with their full-size intermediate in HBM.
//
//   forward  (mi355x_ffn_fwd):        h   = y @ W1^T + b1                              bf16 [M, dff]  (all backward needs)
//                                     out = x + alpha * drop_res( drop_in(swish(h)) @ W2^T + b2 )   f32 [M, 512]
//   backward (mi355x_ffn_bwd_dgrad):  g   = df @ W2                                    (gradient w.r.t. the activated hidden)
//                                     dh  = g * dropmask_in * swish'(h)                bf16 [M, dff]  (wgrad operand of W1)
//                                     a   = drop_in(swish(h))                          bf16 [M, dff]  (wgrad operand of W2, recomputed)
//                                     dy  = dh @ W1                                    bf16 [M, 512]
//
// Replaces on the reference path: ConformerFeedForward.forward, parts/submodules/conformer_modules.py:366-387 (Linear ->
// Swish -> Dropout -> Linear) together with the macaron residual `residual + dropout(ff(x)) * fc_factor` of :174-181 / :209-215,
// and its autograd backward.  Under bf16 autocast the reference rounds the hidden pre-activation to bf16 before the Swish
// (linear1's output IS a bf16 tensor); so does this kernel, and forward and backward therefore see the same bits.
//
// Structure.  A workgroup owns 64 tokens (M = 16 032 -> 251 workgroups for 256 CUs) and walks d_ff in chunks of 64:
//   phase 1   hT_c[64 dff x 64 tok] = W1_c[64 x 512] . y^T          (K = 512, the token operand lives in REGISTERS)
//   transform h_c -> global, act_c = drop(swish(h_c)) -> LDS        (8 consecutive columns per thread: the layout of the mask hash)
//   phase 2   out[64 tok x 512]    += act_c[64 x 64] . W2_c^T       (K = 64 per chunk, accumulators live in REGISTERS)
// act never reaches HBM; the Swish / dropout VALU work sits between MFMA phases instead of behind a drained K loop.
// The eight waves are SPECIALISED: waves 0-3 run phase 1 (they hold the 64 x 512 token operand as MFMA B fragments, 128 VGPRs),
// waves 4-7 run phase 2 (they hold the 64 x 512 f32 accumulators, 128 VGPRs) -- no wave needs both, which is what makes the
// block fit in 256 registers -- and every SIMD hosts one wave of each kind, so the matrix pipe alternates between them while the
// other one reads LDS / does VALU work.  Phase 2 trails phase 1 by one and a half chunks (hand-off through two small LDS buffers).
// Weights stream through a ring of four 32-KiB LDS slots filled by LDS-DMA three steps ahead (counted vmcnt, raw s_barrier);
// a step = a quarter chunk = 8 MFMAs per wave.  The weight images are PRE-PACKED (nemo_amd/packing.py: `ffn_k512` / `ffn_kchunk`)
// in exactly the order the steps consume them, fragment-major, so that (a) every LDS-DMA wave-instruction copies 1 KiB of
// contiguous global memory and (b) every MFMA fragment read is one conflict-free 1-KiB ds_read_b128 burst without a swizzle.
//
// Roofline of this design: per workgroup 4 MiB of weights cross the 64 B/clk vector-memory return path
// (65.5 k cycles) and 8192 MFMAs occupy each SIMD for 65.5 k cycles: at 64 tokens per workgroup the two are equal, so the kernel
// is bound by the L2 -> LDS fill path and the matrix pipe at the same time; HBM traffic is the compulsory y / h / x / out only.
#include <stdlib.h>
#include "common.h"
#include "mi355x_asr.h"

#define FF_D 512
#define FF_BM 64
#define FF_NC 64
#define FF_SLOT 32768
#define FF_WPART 16384                  // first half of a slot: phase-1 weights; second half: phase-2 weights
#define FF_RING (4 * FF_SLOT)           // 131072
#define FF_HBUF FF_RING                 // 8 KiB: [64 tok][64 dff] bf16, 16-B chunks XOR-swizzled by the row
#define FF_HIN (FF_HBUF + 8192)         // 8 KiB (backward only): the stored pre-activation chunk, same image, filled by LDS-DMA
#define FF_ACT (FF_HBUF + 8192)         // forward: 2 x 8 KiB activated chunk, fragment-major [nt][ks][32 tok][16 k]
#define FF_ACT_B (FF_HIN + 8192)        // backward: same, behind the h-in buffer
#define FF_LDS_FWD (FF_ACT + 16384 + 8192)    // + 8 KiB b1 table (dff <= 2048)            = 163840
#define FF_LDS_BWD (FF_ACT_B + 16384)         //                                           = 163840
#define FF_B1 (FF_ACT + 16384)

typedef __attribute__((address_space(3))) void ff_lds_void_t;
typedef const __attribute__((address_space(1))) void ff_glb_void_t;

struct FfnP {
  const bf16_t* tok;    // forward: y = LN(x) [M, 512] bf16 ; backward: df [M, 512] bf16   (row pitch ld_tok)
  const bf16_t* wa;     // phase-1 weights, packed `ffn_k512`  : forward W1, backward W2^T   ([dff][512] logical)
  const bf16_t* wb;     // phase-2 weights, packed `ffn_kchunk`: forward W2, backward W1^T   ([512][dff] logical)
  const float* b1; const float* b2;
  const float* resid;   // forward: x f32 [M, 512]
  void* out;            // forward: f32 [M, 512] ; backward: dy bf16 [M, 512]
  bf16_t* h;            // [M, dff] bf16: written by forward, read by backward
  bf16_t* dh;           // backward out [M, dff]
  bf16_t* act;          // backward out [M, dff]
  long long ld_tok, ld_h, ld_out, ld_res;
  int M, dff;
  float alpha;
  DropCfg d_in, d_res;
  unsigned* trace;   // -DFFN_TRACE builds only: per-segment cycle sums of three workgroups (tools/ffn_trace.py)
  int dbg;   // ablation bits for tools/ffn_bench.py (MI355X_FFN_DBG): 1 no LDS-DMA, 2 no phase-1 MFMAs, 4 no phase-2 MFMAs,
             // 8 no transform, 16 no fragment reads, 32 no step barriers, 64 no steps at all (results are garbage with any bit set)
};

__device__ __forceinline__ void ff_wait_vm(int n) {  // n is uniform
  if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// LDS reads the compiler must not guard with `s_waitcnt vmcnt(0)`: with an LDS-DMA in flight hipcc drains the vector-memory
// queue in front of every ds_read whose address it cannot separate from the DMA destinations (measured in the ISA: the bias
// table read of the hand-off did, i.e. the whole 3-step prefetch was drained once per chunk).  Self-contained (issue + wait).
typedef __attribute__((address_space(3))) const char ff_lds_cchar_t;
__device__ __forceinline__ uint32_t ff_lds_addr(const void* p) { return (uint32_t)(uintptr_t)(ff_lds_cchar_t*)p; }
__device__ __forceinline__ f32x4 ff_lds_rd128_sync(uint32_t addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

__device__ __forceinline__ void ff_lds_rd128x4_sync(uint32_t addr, f32x4 (&v)[4]) {   // addr, +32, +64, +96 bytes
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\tds_read_b128 %2, %4 offset:64\n\t"
               "ds_read_b128 %3, %4 offset:96\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(addr) : "memory");
}

// One LDS-DMA piece: 64 lanes x 16 B = 1 KiB of contiguous global memory -> 1 KiB of LDS at `dst` (wave-uniform)
__device__ __forceinline__ void ff_dma(const void* src_lane, char* dst_uniform) {
  __builtin_amdgcn_global_load_lds((ff_glb_void_t*)src_lane, (ff_lds_void_t*)dst_uniform, 16, 0, 0);
}

#ifdef FFN_ABLATE
#define FF_DBG(b) (p.dbg & (b))
#else
#define FF_DBG(b) false
#endif
// -DFFN_TRACE: s_memtime at the segment boundaries of every step, summed per segment in scalar registers (wave 0 = a phase-1
// wave, wave 4 = a phase-2 wave) and written out at the end -- where inside a step the cycles go.  Each timestamp is followed by
// lgkmcnt(0) (s_memtime returns through the scalar cache), which is why the boundaries sit where no LDS read is in flight.
#ifdef FFN_TRACE
#define FF_TS(k)                                                                                  \
  {                                                                                               \
    unsigned long long t_;                                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
    const unsigned now_ = (unsigned)t_;                                                           \
    tsum[k] += now_ - tlast;                                                                      \
    tlast = now_;                                                                                 \
  }
#else
#define FF_TS(k)
#endif
template <int V> struct FfIC { static constexpr int value = V; };
// per-step work flags (compile-time: a step of the steady state contains no branch -- measured: every uniform branch costs a step
// ~40 cycles of instruction-fetch bubble, and the first version of this loop spent ~600 of its ~2000 cycles per step on them)
enum { FF_F_P1 = 1, FF_F_HAND = 2, FF_F_XFORM = 4, FF_F_P2 = 8, FF_F_ISSUE = 16, FF_F_W4 = 64, FF_F_W0 = 128 };

template <bool BWD>
__global__ __launch_bounds__(512) void ffn_fused_kernel(FfnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  drop_resolve(p.d_in);
  drop_resolve(p.d_res);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, hh = lane >> 5;
  const int m0 = blockIdx.x * FF_BM;
  const int nc = p.dff / FF_NC;     // >= 2
  // steps s = 4 P + J, s < 4 (nc + 1) + 3: phase 1 works on chunk P, the hand-off (J = 0) and the transform (J = 1) on chunk P - 1;
  // phase 2 READS the fragments of chunk P - 2 (J = 0, 1: quarters 2, 3) / P - 1 (J = 2, 3: quarters 0, 1) and multiplies them
  // one step later
  constexpr int ACT = BWD ? FF_ACT_B : FF_ACT;
#ifdef FFN_TRACE
  unsigned tsum[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, tlast = 0u;
#endif

  if (!BWD) {
    float* b1s = reinterpret_cast<float*>(smem + FF_B1);
    for (int i = tid; i < p.dff; i += 512) b1s[i] = p.b1 ? p.b1[i] : 0.f;
  }

  // ---- LDS-DMA sources of this wave: waves 0-3 bring the phase-1 half of every slot (16 pieces of 1 KiB, 4 each),
  // waves 4-7 the phase-2 half (each its OWN 4-KiB slab: it is the only reader).  Both packed images are linear in the step.
  const bool is_p1 = wave < 4;
  const char* dsrc = is_p1 ? (const char*)p.wa + (wave * 4) * 1024 + lane * 16
                           : (const char*)p.wb + ((wave - 4) * 4) * 1024 + lane * 16;
  const int dst_off = is_p1 ? wave * 4096 : FF_WPART + (wave - 4) * 4096;
  const int last_a = 4 * nc - 1;
  // the four pieces of image step t into ring slot `slot`; out-of-range steps re-read a valid one (nobody reads the result):
  // every wave always has exactly 4 pieces per step in flight, which is what the counted waits assume
  auto issue = [&](int t, int slot) {
    if (FF_DBG(1)) return;
    t = t < 0 ? 0 : (t > last_a ? last_a : t);
    const char* src = dsrc + (long long)t * FF_WPART;
    char* dst = smem + slot * FF_SLOT + dst_off;
#pragma unroll
    for (int i = 0; i < 4; ++i) ff_dma(src + i * 1024, dst + i * 1024);
  };
  // backward: the stored pre-activation chunk c -> FF_HIN by DMA (8 pieces of 1 KiB = 8 token rows of 128 B each; one per wave).
  // lane -> (row = 8 w + lane / 8, 16-B chunk lane % 8); the LDS image is lane-linear, so the chunk swizzle is applied to the SOURCE.
  const char* hin_src = nullptr;
  if (BWD) {
    const int row = wave * 8 + (lane >> 3), ch = (lane & 7) ^ (row & 7);
    int m = m0 + row; m = m < p.M ? m : p.M - 1;
    hin_src = (const char*)(p.h + (long long)m * p.ld_h) + ch * 16;
  }
  auto issue_hin = [&](int c) { ff_dma(hin_src + c * (FF_NC * 2), smem + FF_HIN + wave * 1024); };

  // ---- transform of chunk c (all 512 threads; thread = token row tid / 8, 8 consecutive d_ff columns):
  // forward:  h (bf16, from FF_HBUF) -> global h ; act = drop(swish(h)) -> LDS fragment-major
  // backward: g (bf16, from FF_HBUF), h (from FF_HIN) -> dh = g * mask * swish'(h) -> global dh + LDS ; a = drop(swish(h)) -> global act
  const int x_row = tid >> 3, x_ch = tid & 7;
  const int x_m = m0 + x_row;
  const bool x_ok = x_m < p.M;
  const uint32_t x_rd = ff_lds_addr(smem + FF_HBUF) + (uint32_t)(x_row * 128 + ((x_ch ^ (x_row & 7)) * 16));
  char* const x_wr = smem + ACT + x_row * 128 + ((x_ch ^ ((x_row >> 1) & 7)) * 16);
  const long long x_gi = (long long)x_m * p.ld_h + x_ch * 8;
  const uint32_t x_didx = (uint32_t)x_m * (uint32_t)p.dff + (uint32_t)(x_ch * 8);
  auto transform = [&](int c) {
    if (FF_DBG(8)) return;
    const u32x4 raw = __builtin_bit_cast(u32x4, ff_lds_rd128_sync(x_rd));
    u32x4 hr = raw;
    if (BWD) hr = __builtin_bit_cast(u32x4, ff_lds_rd128_sync(x_rd + (FF_HIN - FF_HBUF)));
    const long long gi = x_gi + c * FF_NC;
    if (!BWD && x_ok) *reinterpret_cast<u32x4*>(p.h + gi) = raw;
    // the 8 masks of this thread's index group, generated on the fly (same stream as drop_mask8: one hash per group of 8
    // consecutive indices seeds a xorshift32 sequence) -- streaming form: a pair of elements is finished before the next is touched,
    // which keeps the live registers of this VALU block small (the kernel sits at the 256-register limit)
    const bool dr = p.d_in.threshold != 0u;
    uint32_t st = dr ? drop_group_seed(p.d_in, (x_didx + (uint32_t)(c * FF_NC)) >> 3) : 1u;
    u32x4 o, oa;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      float r2[2], a2[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        st = xorshift32(st);
        const float m = dr ? (st >= p.d_in.threshold ? p.d_in.scale : 0.f) : 1.f;
        const float hv = __uint_as_float(e ? (hr[w] & 0xffff0000u) : (hr[w] << 16));
        const float sg = sigmoidf_(hv);
        if (!BWD) {
          r2[e] = hv * sg * m;
        } else {
          const float gv = __uint_as_float(e ? (raw[w] & 0xffff0000u) : (raw[w] << 16));
          a2[e] = hv * sg * m;
          r2[e] = gv * m * (sg * (1.f + hv * (1.f - sg)));
        }
      }
      o[w] = pack_bf2(r2[0], r2[1]);
      if (BWD) oa[w] = pack_bf2(a2[0], a2[1]);
    }
    if (BWD && x_ok) {
      *reinterpret_cast<u32x4*>(p.dh + gi) = o;
      *reinterpret_cast<u32x4*>(p.act + gi) = oa;
    }
    // the phase-2 token operand: row-major [64][64] bf16, chunks swizzled (see the phase-2 fragment reads)
    *reinterpret_cast<u32x4*>(x_wr + (c & 1) * 8192) = o;
  };
  auto step_end = [&](auto Fc) {
    constexpr int F = decltype(Fc)::value;
    // (MFMAs are register-only instructions: without the scheduling barriers hipcc hoists the NEXT step's MFMAs above the
    //  s_barrier, right behind the fragment reads they consume -- which puts the read -> wait -> multiply chain back together)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (F & FF_F_W0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (F & FF_F_W4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    FF_TS(5)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!FF_DBG(32)) __builtin_amdgcn_s_barrier();
    FF_TS(6)
    __builtin_amdgcn_sched_barrier(0);
  };

  auto issue1 = [&](int t, int slot, int i) {   // one of the four pieces
    if (FF_DBG(1)) return;
    t = t < 0 ? 0 : (t > last_a ? last_a : t);
    ff_dma(dsrc + (long long)t * FF_WPART + i * 1024, smem + slot * FF_SLOT + dst_off + i * 1024);
  };

  // prologue: four steps in flight (phase-2 waves: clamped dummies, so that every wave counts the same pieces)
  issue(is_p1 ? 0 : -6, 0);
  issue(is_p1 ? 1 : -5, 1);
  issue(is_p1 ? 2 : -4, 2);
  issue(is_p1 ? 3 : -3, 3);

  // ---- The step loops.  Measured with -DFFN_TRACE (tools/ffn_trace.py, profiles/r4_ffn_fused.md): a wave's MFMA issue BLOCKS it
  // while the SIMD's matrix pipe works through both waves' queues, an LDS-DMA wave-instruction blocks it ~90 cycles while all
  // eight waves feed the one vector-memory pipe (32 pieces x 16 cycles = the 512-cycle fill floor of a step), a fragment-read
  // burst ~250 cycles -- and run one after the other these add up to three times the 512 MFMA cycles of a step.  So every wave
  // is SOFTWARE-PIPELINED by one step: the fragments a step multiplies were read during the previous step, and the memory
  // instructions of a step (one fragment read per MFMA, into the register that MFMA just released; one LDS-DMA piece per other
  // MFMA) are interleaved with its MFMAs one by one -- while the wave waits for the matrix pipe its loads are already under way.
  constexpr int IS = FF_F_ISSUE, XF = FF_F_XFORM, MM = FF_F_P2, RD = FF_F_P1, HD = FF_F_HAND;
  if (is_p1) {
    // =========================================================================== phase-1 waves
    // MFMAs of image step s (chunk P = s / 4, k-steps 8 J .. 8 J + 7) at step s; its fragments are read at step s - 1 from ring
    // slot s % 4; the pieces of step s + 4 are issued at step s into slot s % 4 (whose fragments went to registers a step ago).
    const int mt = wave >> 1, nt = wave & 1;
    bf16x8 xf[32];   // B fragments of this wave's 32 tokens, all of K = 512: lane (token lr, half hh) holds k = 16 ks + 8 hh .. + 7
    {
      int tokr = m0 + nt * 32 + lr;
      tokr = tokr < p.M ? tokr : p.M - 1;
      const bf16_t* src = p.tok + (long long)tokr * p.ld_tok + hh * 8;
#pragma unroll
      for (int ks = 0; ks < 32; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(src + ks * 16);
    }
    f32x16 acc;   // (one chain: the other wave of the SIMD and the interleaved loads space its MFMAs anyway; 16 registers matter here)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int trow = nt * 32 + lr;   // token row of this lane inside the tile
    const uint32_t b1_addr = ff_lds_addr(smem + FF_B1) + (uint32_t)((mt * 32 + 4 * hh) * 4);
    char* const hb_wr = smem + FF_HBUF + trow * 128 + hh * 8;
    const char* const w_rd = smem + mt * 1024 + lane * 16;   // fragment (ks, mt) = 1 KiB, linear in the lane index
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");   // image steps 0 and 1 landed
    __builtin_amdgcn_s_barrier();
    bf16x8 a[8];     // A fragments (this wave's 32 weight rows x 16 k each) of the step about to be multiplied
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) a[ks] = *reinterpret_cast<const bf16x8*>(w_rd + ks * 2048);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // slot 0 may be refilled
#ifdef FFN_TRACE
    { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); tlast = (unsigned)t_; }
#endif

    auto step = [&](auto Jc, auto Fc, const int P) {
      constexpr int J = decltype(Jc)::value, F = decltype(Fc)::value;
      FF_TS(0)
      // the transform of chunk P - 1 goes FIRST here and LAST in the phase-2 waves: its ~500 VALU cycles per wave then run beside
      // the other wave's MFMAs on both sides (done by both kinds at the same time it was a 1100-cycle hole in the matrix pipe)
      if constexpr (J == 1 && (F & XF)) transform(P - 1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (BWD && J == 2 && (F & MM)) issue_hin(P);   // BEFORE the ring pieces: landed one step later, read at 4 (P + 1) + 1
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if constexpr (F & MM) {
          if (!FF_DBG(2)) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], xf[J * 8 + ks], acc, 0, 0, 0);
          } else {
            asm volatile("" :: "v"(a[ks]));
          }
        }
        if constexpr (F & RD) {
          if (!FF_DBG(16)) a[ks] = *reinterpret_cast<const bf16x8*>(w_rd + ((J + 1) & 3) * FF_SLOT + ks * 2048);
        }
        if constexpr (F & IS) {
          if (ks & 1) issue1(4 * P + J + 4, J, ks >> 1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      FF_TS(1)
      if constexpr (J == 3 && (F & MM)) {   // the chunk is complete: (+ bias) -> bf16 -> FF_HBUF for the transform two steps on
        const int cb = P * FF_NC;             // (FF_HBUF was last read by the transform of chunk P - 1, at step 4 P + 1)
        f32x4 bb[4];
        if (!BWD) ff_lds_rd128x4_sync(b1_addr + (uint32_t)(cb * 4), bb);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 b = {0.f, 0.f, 0.f, 0.f};
          if (!BWD) b = bb[g];
          const u32x2 o = {pack_bf2(acc[4 * g] + b[0], acc[4 * g + 1] + b[1]), pack_bf2(acc[4 * g + 2] + b[2], acc[4 * g + 3] + b[3])};
          *reinterpret_cast<u32x2*>(hb_wr + (((mt * 4 + g) ^ (trow & 7)) * 16)) = o;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      }
      FF_TS(2)
      FF_TS(3)
      FF_TS(4)
      step_end(Fc);
    };
    if (!FF_DBG(64)) {
    step(FfIC<0>{}, FfIC<MM | RD | IS>{}, 0); step(FfIC<1>{}, FfIC<MM | RD | IS>{}, 0);
    step(FfIC<2>{}, FfIC<MM | RD | IS>{}, 0); step(FfIC<3>{}, FfIC<MM | RD | IS>{}, 0);
    for (int P = 1; P < nc - 1; ++P) {
      step(FfIC<0>{}, FfIC<MM | RD | IS>{}, P); step(FfIC<1>{}, FfIC<MM | RD | IS | XF>{}, P);
      step(FfIC<2>{}, FfIC<MM | RD | IS>{}, P); step(FfIC<3>{}, FfIC<MM | RD | IS>{}, P);
    }
    step(FfIC<0>{}, FfIC<MM | RD | IS>{}, nc - 1); step(FfIC<1>{}, FfIC<MM | RD | IS | XF>{}, nc - 1);
    step(FfIC<2>{}, FfIC<MM | RD | IS>{}, nc - 1); step(FfIC<3>{}, FfIC<MM | IS>{}, nc - 1);
    step(FfIC<0>{}, FfIC<FF_F_W0>{}, nc); step(FfIC<1>{}, FfIC<XF | FF_F_W0>{}, nc);
    step(FfIC<2>{}, FfIC<FF_F_W0>{}, nc); step(FfIC<3>{}, FfIC<FF_F_W0>{}, nc);
    step(FfIC<0>{}, FfIC<FF_F_W0>{}, nc + 1); step(FfIC<1>{}, FfIC<FF_F_W0>{}, nc + 1); step(FfIC<2>{}, FfIC<FF_F_W0>{}, nc + 1);
    }
  } else {
    // =========================================================================== phase-2 waves
    // MFMAs of (chunk c, quarter j) at step 4 c + 7 + j; their fragments are read at step 4 c + 6 + j (the transform of chunk c
    // ends with the barrier of step 4 c + 5) from ring slot (step % 4); the weight slab is this wave's OWN DMA (image step s - 3
    // issued at step s into slot (s + 3) % 4, whose previous content was read a step ago).
    const int q = wave - 4;   // output columns 128 q .. 128 q + 127
    f32x16 acc2[2][4];        // [token tile][32-column tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
    // activated chunk: [64 token rows][64 k] bf16 rows of 128 B, 16-B chunks XOR-swizzled by (row / 2) % 8 (conflict-free for the
    // fragment reads below -- a 16-lane group of ds_read_b128 sees 16 distinct (row parity, chunk) pairs -- and for the
    // transform's row-contiguous writes); weight slab: fragments linear in the lane index
    const char* const a_rd = smem + ACT + lr * 128;
    const int a_sw = (lr >> 1) & 7;
    const char* const w_rd = smem + FF_WPART + q * 4096 + lane * 16;
    bf16x8 fa[2], fb[4];      // fragments of the NEXT step's MFMAs
#pragma unroll
    for (int e = 0; e < 8; ++e) fa[0][e] = (__bf16)0.f;
    fa[1] = fa[0]; fb[0] = fa[0]; fb[1] = fa[0]; fb[2] = fa[0]; fb[3] = fa[0];
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
#ifdef FFN_TRACE
    { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); tlast = (unsigned)t_; }
#endif

    // flags: MM = this step multiplies (fragments were read last step); RD = this step reads the fragments of the next one;
    // HD (re-used) = chunk P exists (backward: its stored pre-activations are fetched by all waves)
    auto step = [&](auto Jc, auto Fc, const int P) {
      constexpr int J = decltype(Jc)::value, F = decltype(Fc)::value;
      constexpr int jn = (J + 2) & 3;
      const int cn = J < 2 ? P - 2 : P - 1;
      const char* asl = a_rd + (cn & 1) * 8192 + (((jn * 2 + hh) ^ a_sw) * 16);
      const char* wsl = w_rd + J * FF_SLOT;
      FF_TS(0)
      if constexpr (BWD && J == 2 && (F & HD)) issue_hin(P);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if constexpr (F & MM) {
          if (!FF_DBG(4)) acc2[k >> 2][k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[k >> 2], fb[k & 3], acc2[k >> 2][k & 3], 0, 0, 0);
          else asm volatile("" :: "v"(fa[k >> 2]), "v"(fb[k & 3]));
        }
        if constexpr (F & IS) {
          if (k < 4) issue1(4 * P + J + 3 - 6, (J + 3) & 3, k);
        }
        if constexpr (F & RD) {   // each fragment register is re-loaded behind the last MFMA that reads it
          if (!FF_DBG(16)) {
            if (k == 3) fa[0] = *reinterpret_cast<const bf16x8*>(asl);
            if (k >= 4) fb[k - 4] = *reinterpret_cast<const bf16x8*>(wsl + (k - 4) * 1024);
            if (k == 7) fa[1] = *reinterpret_cast<const bf16x8*>(asl + 4096);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      FF_TS(1)
      if constexpr (J == 1 && (F & XF)) transform(P - 1);
      FF_TS(2)
      FF_TS(3)
      FF_TS(4)
      step_end(Fc);
    };
    if (!FF_DBG(64)) {
    step(FfIC<0>{}, FfIC<IS | HD>{}, 0); step(FfIC<1>{}, FfIC<IS | HD>{}, 0);
    step(FfIC<2>{}, FfIC<IS | HD>{}, 0); step(FfIC<3>{}, FfIC<IS | HD>{}, 0);
    step(FfIC<0>{}, FfIC<IS | HD>{}, 1);
    step(FfIC<1>{}, FfIC<IS | HD | XF>{}, 1);
    step(FfIC<2>{}, FfIC<IS | HD | RD>{}, 1);
    step(FfIC<3>{}, FfIC<IS | HD | RD | MM>{}, 1);
    for (int P = 2; P < nc; ++P) {
      step(FfIC<0>{}, FfIC<IS | HD | RD | MM>{}, P); step(FfIC<1>{}, FfIC<IS | HD | RD | MM | XF>{}, P);
      step(FfIC<2>{}, FfIC<IS | HD | RD | MM>{}, P); step(FfIC<3>{}, FfIC<IS | HD | RD | MM>{}, P);
    }
    step(FfIC<0>{}, FfIC<IS | RD | MM>{}, nc);
    step(FfIC<1>{}, FfIC<IS | RD | MM | XF>{}, nc);
    step(FfIC<2>{}, FfIC<IS | RD | MM>{}, nc);
    step(FfIC<3>{}, FfIC<FF_F_W4 | RD | MM>{}, nc);
    step(FfIC<0>{}, FfIC<FF_F_W0 | RD | MM>{}, nc + 1);
    step(FfIC<1>{}, FfIC<FF_F_W0 | RD | MM>{}, nc + 1);
    step(FfIC<2>{}, FfIC<FF_F_W0 | MM>{}, nc + 1);
    }

    // accumulators -> f32 tile image in LDS ([64][516]); C/D layout of the 32x32 MFMA: column = lane & 31,
    // row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float* sC = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sC[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * (FF_D + 4) + q * 128 + j * 32 + lr] = acc2[i][j][r];
  }
#ifdef FFN_TRACE
  if (p.trace && lane == 0 && (wave == 0 || wave == 4)) {
    const int b = blockIdx.x;
    const int slot = b == 0 ? 0 : (b == (int)gridDim.x / 2 ? 1 : (b == (int)gridDim.x - 1 ? 2 : -1));
    if (slot >= 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) p.trace[(slot * 2 + (wave >> 2)) * 8 + k] = tsum[k];
    }
  }
#endif
  // ---- output pass: thread = 8 consecutive columns (tid % 64) of rows tid / 64 + 8 it.  The residual rows are requested
  // BEFORE the barrier (their memory latency passes while the phase-2 waves move the accumulators through LDS).
  const int c8 = (tid & 63) * 8;
  float4 res[BWD ? 1 : 8][2];
  float b8[8];
  if (!BWD) {
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 ba = p.b2 ? *reinterpret_cast<const float4*>(p.b2 + c8) : z4;
    const float4 bb = p.b2 ? *reinterpret_cast<const float4*>(p.b2 + c8 + 4) : z4;
    b8[0] = ba.x; b8[1] = ba.y; b8[2] = ba.z; b8[3] = ba.w; b8[4] = bb.x; b8[5] = bb.y; b8[6] = bb.z; b8[7] = bb.w;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      int m = m0 + (tid >> 6) + 8 * it;
      m = m < p.M ? m : p.M - 1;
      const float* rp = p.resid + (long long)m * p.ld_res + c8;
      res[it][0] = *reinterpret_cast<const float4*>(rp);
      res[it][1] = *reinterpret_cast<const float4*>(rp + 4);
    }
  }
  __syncthreads();
  {
    const float* sC = reinterpret_cast<const float*>(smem);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int rl = (tid >> 6) + 8 * it;
      const int m = m0 + rl;
      const float4 s0 = *reinterpret_cast<const float4*>(sC + rl * (FF_D + 4) + c8);
      const float4 s1 = *reinterpret_cast<const float4*>(sC + rl * (FF_D + 4) + c8 + 4);
      float v[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      if (m >= p.M) continue;
      if (!BWD) {
        float dm[8];
        drop_mask8(p.d_res, (uint32_t)m * (uint32_t)FF_D + (uint32_t)c8, dm);
        const float4 r0 = res[it][0], r1 = res[it][1];
        const float rr[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = rr[j] + p.alpha * (v[j] + b8[j]) * dm[j];
        float* op = (float*)p.out + (long long)m * p.ld_out + c8;
        *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        *reinterpret_cast<u32x4*>((bf16_t*)p.out + (long long)m * p.ld_out + c8) =
            u32x4{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
      }
    }
  }
}

// ================================================================================================= weight images
// The two image orders of the fused kernels, written straight from the fp32 master weights (one 64 x 64 source tile per
// workgroup through LDS: coalesced 256-B reads of the source, 16-B bf16 stores of whole MFMA fragment lanes).  A FRAGMENT is the
// 1 KiB a wave reads with ONE ds_read_b128: 32 rows x 16 k, lane l = lr + 32 hh holds row lr, k = 8 hh .. 8 hh + 7 -- stored
// [hh][lr][8], i.e. linear in the lane index (the first version stored [lr][hh][8]: every fragment read was a 2-way bank
// conflict, SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE, profiles/r4_ffn_fused.md).
//   "k512"   image of a logical A [dff][512]:  frag (c = row / 64, k16 = k / 16, mt = (row % 64) / 32) at ((c * 32 + k16) * 2 + mt) KiB
//   "kchunk" image of a logical B [512][dff]:  frag (t = k / 16, q = row / 128, mt4 = (row % 128) / 32) at ((t * 4 + q) * 4 + mt4) KiB
// W1 [dff][512] gives k512(W1) (forward phase 1) and kchunk(W1^T) (backward phase 2); W2 [512][dff] gives kchunk(W2) (forward
// phase 2) and k512(W2^T) (backward phase 1).
struct FfnPackEntry { const float* src; bf16_t* k512; bf16_t* kchunk; int dff; int is_w2; };
__global__ __launch_bounds__(256) void ffn_pack_kernel(const FfnPackEntry* __restrict__ tab) {
  __shared__ float tile[64][65];
  const FfnPackEntry e = tab[blockIdx.y];
  const int C = e.is_w2 ? e.dff : FF_D;             // source row length
  const int tiles_c = C / 64;
  const int tr = blockIdx.x / tiles_c, tc = blockIdx.x - tr * tiles_c;
  if (tr * 64 >= (e.is_w2 ? FF_D : e.dff)) return;
  const float* src = e.src + (long long)tr * 64 * C + tc * 64;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int qd = threadIdx.x + k * 256;
    const int f4 = (qd & 15) * 4, rr = qd >> 4;
    const float4 v = *reinterpret_cast<const float4*>(src + (long long)rr * C + f4);
    tile[rr][f4] = v.x; tile[rr][f4 + 1] = v.y; tile[rr][f4 + 2] = v.z; tile[rr][f4 + 3] = v.w;
  }
  __syncthreads();
  // dff-side origin X0 and 512-side origin Y0 of this tile
  const int X0 = e.is_w2 ? tc * 64 : tr * 64, Y0 = e.is_w2 ? tr * 64 : tc * 64;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int pc = threadIdx.x + k * 256;            // piece (p, mtl, hh, lr)
    const int lr = pc & 31, hh = (pc >> 5) & 1, mtl = (pc >> 6) & 1, p4 = pc >> 7;
    float rw[8], cw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      rw[j] = tile[mtl * 32 + lr][p4 * 16 + hh * 8 + j];   // row-wise piece: rows of the tile are the fragment rows
      cw[j] = tile[p4 * 16 + hh * 8 + j][mtl * 32 + lr];   // column-wise piece: columns of the tile are the fragment rows
    }
    const u32x4 rv = {pack_bf2(rw[0], rw[1]), pack_bf2(rw[2], rw[3]), pack_bf2(rw[4], rw[5]), pack_bf2(rw[6], rw[7])};
    const u32x4 cv = {pack_bf2(cw[0], cw[1]), pack_bf2(cw[2], cw[3]), pack_bf2(cw[4], cw[5]), pack_bf2(cw[6], cw[7])};
    // W1: rows = dff side (fragment rows of k512 come from tile ROWS, k from tile columns); kchunk(W1^T) fragment rows = the 512 side
    // W2: rows = 512 side: kchunk(W2) fragment rows come from tile ROWS; k512(W2^T) fragment rows (dff side) from tile COLUMNS
    const long long i512 = ((((long long)(X0 / 64) * 32 + Y0 / 16 + p4) * 2 + mtl) * 2 + hh) * 256 + lr * 8;
    const long long ichk = (((((long long)(X0 / 16 + p4)) * 4 + Y0 / 128) * 4 + (Y0 % 128) / 32 + mtl) * 2 + hh) * 256 + lr * 8;
    if (!e.is_w2) {
      *reinterpret_cast<u32x4*>(e.k512 + i512) = rv;
      *reinterpret_cast<u32x4*>(e.kchunk + ichk) = cv;
    } else {
      *reinterpret_cast<u32x4*>(e.kchunk + ichk) = rv;
      *reinterpret_cast<u32x4*>(e.k512 + i512) = cv;
    }
  }
}
extern "C" int mi355x_ffn_pack(const void* table_dev, int n_entries, int max_dff, void* stream) {
  mi_clear_errors();
  if (!table_dev || n_entries <= 0 || max_dff < 2 * FF_NC || (max_dff % FF_NC) || max_dff > 2048) return MI_ERR_ARG;
  MI_LAUNCH(ffn_pack_kernel, dim3((unsigned)(max_dff / 64 * (FF_D / 64)), (unsigned)n_entries), dim3(256), 0, (hipStream_t)stream,
            (const FfnPackEntry*)table_dev);
  return mi_check_launch();
}

static unsigned* g_ffn_trace = nullptr;
extern "C" int mi355x_ffn_debug_trace(void* dev_buf) {  // (diagnostics; not part of the ABI header) 48 unsigned words
  g_ffn_trace = (unsigned*)dev_buf;
  return 0;
}
static int ffn_dbg() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("MI355X_FFN_DBG"); v = e ? atoi(e) : 0; }
  return v;
}
static int ffn_check(const FfnP& p) {
  if (!p.tok || !p.wa || !p.wb || !p.out || !p.h) return MI_ERR_ARG;
  if (p.M <= 0 || p.dff < 2 * FF_NC || (p.dff % FF_NC) || p.dff > 2048) return MI_ERR_ARG;
  if ((p.ld_tok & 7) || (p.ld_h & 7) || (p.ld_out & 7)) return MI_ERR_ARG;
  if (((uintptr_t)p.tok | (uintptr_t)p.wa | (uintptr_t)p.wb | (uintptr_t)p.out | (uintptr_t)p.h) & 15) return MI_ERR_ARG;
  return MI_OK;
}

extern "C" int mi355x_ffn_fwd(const void* y, long long ldy, const void* w1_packed, const void* b1, const void* w2_packed,
                              const void* b2, const void* x_resid, long long ldx, void* h, long long ldh, void* out, long long ldo,
                              int M, int d_model, int d_ff, float alpha, unsigned drop_in_key, unsigned drop_in_threshold,
                              float drop_in_scale, unsigned drop_res_key, unsigned drop_res_threshold, float drop_res_scale,
                              void* stream) {
  mi_clear_errors();
  if (d_model != FF_D || !x_resid || (ldx & 3) || ((uintptr_t)x_resid & 15)) return MI_ERR_ARG;
  FfnP p = {};
  p.tok = (const bf16_t*)y; p.wa = (const bf16_t*)w1_packed; p.wb = (const bf16_t*)w2_packed;
  p.b1 = (const float*)b1; p.b2 = (const float*)b2; p.resid = (const float*)x_resid;
  p.out = out; p.h = (bf16_t*)h;
  p.ld_tok = ldy; p.ld_h = ldh; p.ld_out = ldo; p.ld_res = ldx;
  p.M = M; p.dff = d_ff; p.alpha = alpha;
  p.d_in = mi_drop(drop_in_key, drop_in_threshold, drop_in_scale);
  p.d_res = mi_drop(drop_res_key, drop_res_threshold, drop_res_scale);
  p.dbg = ffn_dbg();
  p.trace = g_ffn_trace;
  if (int e = ffn_check(p)) return e;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)ffn_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS_FWD);
    hipFuncSetAttribute((const void*)ffn_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS_BWD);
    attr_done = true;
  }
  MI_LAUNCH(ffn_fused_kernel<false>, dim3((M + FF_BM - 1) / FF_BM), dim3(512), FF_LDS_FWD, (hipStream_t)stream, p);
  return mi_check_launch();
}

extern "C" int mi355x_ffn_bwd_dgrad(const void* df, long long lddf, const void* w2t_packed, const void* w1t_packed, const void* h,
                                    long long ldh, void* dh, void* act, void* dy, long long lddy, int M, int d_model, int d_ff,
                                    unsigned drop_in_key, unsigned drop_in_threshold, float drop_in_scale, void* stream) {
  mi_clear_errors();
  if (d_model != FF_D || !dh || !act || (((uintptr_t)dh | (uintptr_t)act) & 15)) return MI_ERR_ARG;
  FfnP p = {};
  p.tok = (const bf16_t*)df; p.wa = (const bf16_t*)w2t_packed; p.wb = (const bf16_t*)w1t_packed;
  p.out = dy; p.h = (bf16_t*)const_cast<void*>(h); p.dh = (bf16_t*)dh; p.act = (bf16_t*)act;
  p.ld_tok = lddf; p.ld_h = ldh; p.ld_out = lddy;
  p.M = M; p.dff = d_ff; p.alpha = 1.f;
  p.d_in = mi_drop(drop_in_key, drop_in_threshold, drop_in_scale);
  p.d_res = mi_drop(0u, 0u, 1.f);
  p.dbg = ffn_dbg();
  p.trace = g_ffn_trace;
  if (int e = ffn_check(p)) return e;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)ffn_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS_FWD);
    hipFuncSetAttribute((const void*)ffn_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS_BWD);
    attr_done = true;
  }
  MI_LAUNCH(ffn_fused_kernel<true>, dim3((M + FF_BM - 1) / FF_BM), dim3(512), FF_LDS_BWD, (hipStream_t)stream, p);
  return mi_check_launch();
}
