// CTC loss + gradient w.r.t. log-probabilities, one workgroup per utterance.
//   phase 1: alpha (waves 0-1) and beta (waves 2-3) recursions run CONCURRENTLY in log space over the blank-extended
//            label sequence (S = 2U+1 states, previous time-step row double-buffered in LDS, one barrier per step);
//            full lattices are written to a workspace [B, T, S] each.
//   phase 2: grad[t,c] = -exp(alpha+beta - logp + nll) summed over the states carrying class c, scaled by grad_scale
//            (1/B for 'mean_batch'); one wave per time-step, per-wave class accumulators in LDS (ds_add_f32).
// zero_infinity: an infeasible utterance gets loss 0 and zero gradient.
//
// Replaces on the reference path: torch.nn.functional.ctc_loss forward+backward called by
//   nemo/collections/asr/losses/ctc.py:68-82 (CTCLoss(blank=V, reduction='none', zero_infinity=True) + mean_batch).
#include "common.h"
#include "mi355x_asr.h"

#define NEGINF (-INFINITY)

__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(a, fmaxf(b, c));
  if (m == NEGINF) return NEGINF;
  return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

// logp [B, Tmax, C] f32 ; targets [B, Umax] i64 ; alpha/beta workspaces [B, Tmax, Smax]; grad [B, Tmax, C]
__global__ __launch_bounds__(256) void ctc_kernel(const float* __restrict__ logp, const long long* __restrict__ targets,
                                                  const long long* __restrict__ in_len, const long long* __restrict__ tgt_len,
                                                  float* __restrict__ alpha_ws, float* __restrict__ beta_ws,
                                                  float* __restrict__ nll_out, int Tmax, int C, int Umax, int Smax, int blank,
                                                  int zero_infinity) {
  extern __shared__ float sm[];
  int* ext = (int*)sm;                 // [Smax]
  float* prev_a = sm + Smax;           // [2][Smax]
  float* prev_b = prev_a + 2 * Smax;   // [2][Smax]

  const int b = blockIdx.x;
  const int T = (int)min((long long)Tmax, in_len[b]);
  const int U = (int)min((long long)Umax, tgt_len[b]);
  const int S = 2 * U + 1;
  const float* lp = logp + (long long)b * Tmax * C;
  float* aw = alpha_ws + (long long)b * Tmax * Smax;
  float* bw = beta_ws + (long long)b * Tmax * Smax;
  const int tid = threadIdx.x;

  for (int s = tid; s < S; s += 256) ext[s] = (s & 1) ? (int)targets[(long long)b * Umax + (s >> 1)] : blank;
  __syncthreads();

  if (T <= 0) {  // empty input: feasible only for an empty target
    if (tid == 0) nll_out[b] = (U == 0) ? 0.f : (zero_infinity ? 0.f : INFINITY);
    return;
  }

  // ---------------- phase 1: alpha on threads [0,128), beta on threads [128,256)
  const bool is_beta = tid >= 128;
  const int ht = tid & 127;
  float* prev = is_beta ? prev_b : prev_a;
  float* ws = is_beta ? bw : aw;
  // The emission term lp[t, ext[s]] does not depend on the recursion: it is fetched four time-steps ahead into a register
  // ring (a global load per step on the critical path was ~1 us x 501 steps = the whole kernel).  Ring slot = step & 3;
  // the loop is unrolled by 4 so that the slot is a compile-time register.
  const int cls0 = (ht < S) ? ext[ht] : blank;
  auto emit = [&](int step) -> float {
    const int t = is_beta ? (T - 1 - step) : step;
    return (step < T && ht < S) ? lp[(long long)t * C + cls0] : 0.f;
  };
  float ring[4] = {emit(0), emit(1), emit(2), emit(3)};
  for (int step0 = 0; step0 < T; step0 += 4) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int step = step0 + k;
      if (step < T) {  // uniform over the workgroup
        const float e0 = ring[k];
        ring[k] = emit(step + 4);
        const int t = is_beta ? (T - 1 - step) : step;
        float* cur = prev + ((step & 1) ? Smax : 0);
        const float* old = prev + ((step & 1) ? 0 : Smax);
        for (int s = ht; s < S; s += 128) {
          float v;
          const int cls = (s == ht) ? cls0 : ext[s];
          const float e = (s == ht) ? e0 : lp[(long long)t * C + cls];
          if (step == 0) {
            if (!is_beta) v = (s < 2) ? e : NEGINF;
            else v = (s >= S - 2) ? e : NEGINF;
          } else {
            float a = old[s], bb, c = NEGINF;
            if (!is_beta) {
              bb = (s >= 1) ? old[s - 1] : NEGINF;
              if (s >= 2 && cls != blank && cls != ext[s - 2]) c = old[s - 2];
            } else {
              bb = (s + 1 < S) ? old[s + 1] : NEGINF;
              if (s + 2 < S && cls != blank && cls != ext[s + 2]) c = old[s + 2];
            }
            const float m = lse3(a, bb, c);
            v = (m == NEGINF) ? NEGINF : m + e;
          }
          cur[s] = v;
          ws[(long long)t * Smax + s] = v;
        }
        __syncthreads();
      }
    }
  }
  // log-likelihood from the last alpha row (= row (T-1)&1 of prev_a)
  if (tid == 0) {
    const float* last = prev_a + (((T - 1) & 1) ? Smax : 0);
    const float l1 = last[S - 1], l2 = (S > 1) ? last[S - 2] : NEGINF;
    const float m = fmaxf(l1, l2);
    float ll = (m == NEGINF) ? NEGINF : m + logf(expf(l1 - m) + expf(l2 - m));
    float out = -ll;
    if (out == INFINITY && zero_infinity) out = 0.f;
    nll_out[b] = out;
  }
}

// ---------------- phase 2: gradient rows.  Its own launch: the lattice phase is sequential in T and occupies one workgroup per
// utterance (32 CUs at the headline batch), the gradient rows are independent -- grid (time blocks, B), one wave per time-step.
#define CTC_GT 8  // time-steps per workgroup (4 waves x 2)
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ logp, const long long* __restrict__ targets,
                                                       const long long* __restrict__ in_len, const long long* __restrict__ tgt_len,
                                                       const float* __restrict__ alpha_ws, const float* __restrict__ beta_ws,
                                                       float* __restrict__ grad, int Tmax, int C, int Umax, int Smax, int blank,
                                                       float grad_scale) {
  extern __shared__ float acc[];  // [4][C]
  const int b = blockIdx.y;
  const int T = (int)min((long long)Tmax, in_len[b]);
  const int U = (int)min((long long)Umax, tgt_len[b]);
  const int S = 2 * U + 1;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* lp = logp + (long long)b * Tmax * C;
  const float* aw = alpha_ws + (long long)b * Tmax * Smax;
  float nll = INFINITY;  // the un-clamped negative log-likelihood, from the last alpha row (same arithmetic as phase 1)
  if (T > 0) {
    const float l1 = aw[(long long)(T - 1) * Smax + S - 1], l2 = (S > 1) ? aw[(long long)(T - 1) * Smax + S - 2] : NEGINF;
    const float m = fmaxf(l1, l2);
    nll = -((m == NEGINF) ? NEGINF : m + logf(expf(l1 - m) + expf(l2 - m)));
  }
  const float* bw = beta_ws + (long long)b * Tmax * Smax;
  float* g = grad + (long long)b * Tmax * C;
  float* wacc = acc + wave * C;
  const bool feasible = T > 0 && nll != INFINITY;  // infeasible (zero_infinity) / empty: zero rows
  for (int t = blockIdx.x * CTC_GT + wave; t < min(Tmax, (int)(blockIdx.x + 1) * CTC_GT); t += 4) {
    if (t >= T || !feasible) {  // frames beyond the utterance stay zero
      for (int c = lane; c < C; c += 64) g[(long long)t * C + c] = 0.f;
      continue;
    }
    for (int c = lane; c < C; c += 64) wacc[c] = 0.f;
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS zeroing visible before the adds of this wave
    for (int s = lane; s < S; s += 64) {
      const float ab = aw[(long long)t * Smax + s] + bw[(long long)t * Smax + s];
      if (ab != NEGINF) {
        const int cls = (s & 1) ? (int)targets[(long long)b * Umax + (s >> 1)] : blank;
        atomicAdd(&wacc[cls], expf(ab - lp[(long long)t * C + cls] + nll));
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    for (int c = lane; c < C; c += 64) g[(long long)t * C + c] = -grad_scale * wacc[c];
  }
}

extern "C" int mi355x_ctc_loss(const void* logp, const void* targets, const void* in_len, const void* tgt_len, void* alpha_ws,
                               void* beta_ws, void* nll, void* grad, int B, int Tmax, int C, int Umax, int blank,
                               float grad_scale, int zero_infinity, void* stream) {
  mi_clear_errors();
  if (!logp || !targets || !in_len || !tgt_len || !alpha_ws || !beta_ws || !nll) return MI_ERR_ARG;
  if (B <= 0 || Tmax <= 0 || C <= 0 || Umax < 0 || blank < 0 || blank >= C) return MI_ERR_ARG;
  const int Smax = 2 * Umax + 1;
  const size_t shm = sizeof(float) * ((size_t)Smax * 5);
  if (shm > 60 * 1024 || sizeof(float) * 4 * (size_t)C > 60 * 1024) return MI_ERR_ARG;
  MI_LAUNCH(ctc_kernel, dim3(B), dim3(256), shm, (hipStream_t)stream, (const float*)logp, (const long long*)targets,
                     (const long long*)in_len, (const long long*)tgt_len, (float*)alpha_ws, (float*)beta_ws, (float*)nll, Tmax,
                     C, Umax, Smax, blank, zero_infinity);
  if (grad)
    MI_LAUNCH(ctc_grad_kernel, dim3((Tmax + CTC_GT - 1) / CTC_GT, B), dim3(256), sizeof(float) * 4 * (size_t)C,
                       (hipStream_t)stream, (const float*)logp, (const long long*)targets, (const long long*)in_len,
                       (const long long*)tgt_len, (const float*)alpha_ws, (const float*)beta_ws, (float*)grad, Tmax, C, Umax,
                       Smax, blank, grad_scale);
  return mi_check_launch();
}

// ------------------------------------------------------------------------------------------------ greedy CTC decoding
// GreedyCTCInfer._greedy_decode_logprobs (parts/submodules/ctc_greedy_decoding.py:333-361) + the CTC collapse of
// AbstractCTCDecoding.decode_hypothesis (parts/submodules/ctc_decoding.py:545-575), one workgroup per utterance, no host
// round trip: per-frame argmax (first maximum, like torch.max) and its log-probability, score = sum of the log-probs of
// the non-blank frames, tokens = labels with repeats folded and blanks removed (wave ballot + prefix popcount compaction).
__global__ __launch_bounds__(256) void ctc_greedy_kernel(const float* __restrict__ logp, const long long* __restrict__ lens,
                                                         int* __restrict__ tokens, int* __restrict__ out_len,
                                                         float* __restrict__ score, int Tmax, int C, int blank) {
  extern __shared__ int s_lab[];  // [Tmax] labels ; then [Tmax] log-probs (as float)
  float* s_lp = reinterpret_cast<float*>(s_lab + Tmax);
  const int b = blockIdx.x;
  const int T = (int)min((long long)Tmax, lens ? lens[b] : (long long)Tmax);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* lp = logp + (long long)b * Tmax * C;
  for (int t = wave; t < T; t += 4) {
    float best = -INFINITY; int arg = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
      const float v = lp[(long long)t * C + c];
      if (v > best) { best = v; arg = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64); const int oa = __shfl_xor(arg, o, 64);
      if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if (lane == 0) { s_lab[t] = arg; s_lp[t] = best; }
  }
  __syncthreads();
  int* tok = tokens + (long long)b * Tmax;
  if (wave == 0) {
    int base = 0; float sc = 0.f;
    for (int t0 = 0; t0 < T; t0 += 64) {
      const int t = t0 + lane;
      const int cur = t < T ? s_lab[t] : blank;
      const int prev = (t > 0 && t < T) ? s_lab[t - 1] : blank;
      const bool nonblank = t < T && cur != blank;
      const bool keep = nonblank && cur != prev;
      if (nonblank) sc += s_lp[t];
      const unsigned long long m = __ballot(keep);
      const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
      if (keep) tok[pos] = cur;
      base += __popcll(m);
    }
    sc = wave_sum(sc);
    if (lane == 0) { out_len[b] = base; score[b] = sc; }
    for (int t = base + lane; t < Tmax; t += 64) tok[t] = -1;  // padding
  }
}
extern "C" int mi355x_ctc_greedy_decode(const void* logp, const void* lens, void* tokens, void* out_len, void* score, int B,
                                        int Tmax, int C, int blank, void* stream) {
  mi_clear_errors();
  if (!logp || !tokens || !out_len || !score || B <= 0 || Tmax <= 0 || C <= 0 || blank < 0 || blank > C) return MI_ERR_ARG;
  const size_t shm = (size_t)Tmax * 8;
  if (shm > 64 * 1024) return MI_ERR_ARG;
  MI_LAUNCH(ctc_greedy_kernel, dim3(B), dim3(256), shm, (hipStream_t)stream, (const float*)logp, (const long long*)lens,
                     (int*)tokens, (int*)out_len, (float*)score, Tmax, C, blank);
  return mi_check_launch();
}

