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
#include <stdlib.h>
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

// ---------------- phase 1, wave-resident form (round 5).  The kernel above spends its time on one workgroup barrier + an LDS round
// trip per time-step (~0.5 us x 501 steps = 247 us at the headline shapes, on the critical path between forward and backward:
// profiles/r4_roofline_per_kernel.md).  Here a lattice row never leaves the registers of ONE wave: lane l holds the P (blank, label)
// state pairs g = l*P .. l*P+P-1 (states 2g, 2g+1; S <= 128*P) and a step needs exactly one value (alpha) / two values (beta) from the
// neighbouring lane -- a wave shuffle, no barrier, no LDS round trip of the row.  grid (B, 2): workgroup y = 0 walks alpha forward
// in time, y = 1 walks beta backward.  A workgroup is two waves: wave 0 walks, wave 1 gathers the emissions lp[t, ext[s]] of the NEXT
// chunk of CTC_TC(P) time-steps into the other half of a double-buffered LDS image (all of a chunk's loads in flight at once), so
// the walker never waits for global memory; one barrier per chunk.  The lattices are written exactly as the kernel above writes
// them (both include the emission of their own time-step).
// The walker keeps its rows in the BASE-2 log domain (emissions are scaled by log2(e) on their way into LDS, lattice rows by ln 2
// on their way out): log-sum-exp is then the bare hardware v_exp_f32 / v_log_f32 pair -- the sum lies in [1, 3], so none of the
// range handling of expf / logf is needed.
#define CTC_LOG2E 1.4426950408889634f
#define CTC_LN2 0.6931471805599453f
__device__ __forceinline__ float lse2f(float a, float b) {
  const float m = fmaxf(a, b);
  return (m == NEGINF) ? NEGINF : m + __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(a - m) + __builtin_amdgcn_exp2f(b - m));
}
__device__ __forceinline__ float lse3f(float a, float b, float c) {
  const float m = fmaxf(a, fmaxf(b, c));
  return (m == NEGINF) ? NEGINF
                       : m + __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(a - m) + __builtin_amdgcn_exp2f(b - m) + __builtin_amdgcn_exp2f(c - m));
}
template <int P> struct CtcTc { static constexpr int v = P <= 2 ? 64 : 128 / P; };   // time-steps per chunk (<= 64: one blank load per lane)
template <int P, bool BETA>
__device__ __forceinline__ void ctc_wave_walk(const float* __restrict__ lp, const long long* __restrict__ tg, float* __restrict__ ws,
                                              float* __restrict__ nll_out, int T, int U, int C, int Smax, int blank, int zero_infinity,
                                              float (*s_e)[CtcTc<P>::v * (64 * P + 1)]) {
  constexpr int TC = CtcTc<P>::v, LD = 64 * P + 1;   // LDS row of a time-step: [p][lane] label emissions, then the blank emission
  const int lane = threadIdx.x & 63;
  const bool loader = threadIdx.x >= 64;   // wave-uniform
  // per pair: the label's class, whether the skip transition into (alpha) / out of (beta) its label state exists
  int cls[P];
  bool vE[P], vO[P], skip[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int g = lane * P + p;
    vE[p] = g <= U;
    vO[p] = g < U;
    cls[p] = vO[p] ? (int)tg[g] : blank;
    if (!BETA) skip[p] = vO[p] && g >= 1 && cls[p] != (int)tg[g - 1];
    else skip[p] = g + 1 < U && cls[p] != (int)tg[g + 1];
  }
  const int nchunk = (T + TC - 1) / TC;
  auto gather = [&](int c) {   // loader wave: emissions of steps c*TC .. c*TC+TC-1 -> s_e[c & 1]
    float* dst = s_e[c & 1];
    float v[TC][P];
#pragma unroll
    for (int i = 0; i < TC; ++i) {
      int step = c * TC + i;
      step = step < T ? step : T - 1;   // (steps beyond the utterance re-read its last row: no branch, never used)
      const float* row = lp + (long long)(BETA ? (T - 1 - step) : step) * C;
#pragma unroll
      for (int p = 0; p < P; ++p) v[i][p] = row[cls[p]];
    }
    float vb = 0.f;
    if (lane < TC) {
      int step = c * TC + lane;
      step = step < T ? step : T - 1;
      vb = lp[(long long)(BETA ? (T - 1 - step) : step) * C + blank];
    }
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
      for (int p = 0; p < P; ++p) dst[i * LD + p * 64 + lane] = v[i][p] * CTC_LOG2E;
    if (lane < TC) dst[lane * LD + 64 * P] = vb * CTC_LOG2E;
  };
  float E[P], O[P];
#pragma unroll
  for (int p = 0; p < P; ++p) { E[p] = NEGINF; O[p] = NEGINF; }
  if (loader) gather(0);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    if (loader) {
      if (c + 1 < nchunk) gather(c + 1);
    } else {
      const float* src = s_e[c & 1];
      const int nstep = min(TC, T - c * TC);
      // the emissions of step i + 1 are read while step i computes (an LDS read in front of its first use costs the recursion
      // ~100 cycles per step otherwise); row TC - 1 is re-read beyond the chunk (never used)
      float eb_n = src[64 * P], el_n[P];
#pragma unroll
      for (int p = 0; p < P; ++p) el_n[p] = src[p * 64 + lane];
      for (int i = 0; i < nstep; ++i) {
        const int step = c * TC + i;
        const int t = BETA ? (T - 1 - step) : step;
        const float eb = eb_n;
        float el[P];
#pragma unroll
        for (int p = 0; p < P; ++p) el[p] = el_n[p];
        const int i1 = i + 1 < TC ? i + 1 : TC - 1;
        // old values of the neighbouring lane's boundary pair (alpha: its last label state; beta: its first pair)
        float x0, x1 = NEGINF;
        if (!BETA) {
          x0 = __shfl_up(O[P - 1], 1, 64);
          if (lane == 0) x0 = NEGINF;
        } else {
          x0 = __shfl_down(E[0], 1, 64);
          x1 = __shfl_down(O[0], 1, 64);
          if (lane == 63) { x0 = NEGINF; x1 = NEGINF; }
        }
        eb_n = src[i1 * LD + 64 * P];
#pragma unroll
        for (int p = 0; p < P; ++p) el_n[p] = src[i1 * LD + p * 64 + lane];
        float nE[P], nO[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
          float a, cc;
          if (!BETA) {
            // alpha: E_g <- lse(E_g, O_{g-1}) + e_blank ; O_g <- lse(O_g, E_g, skip ? O_{g-1} : -inf) + e_label   (old values on the right)
            const float om1 = (p == 0) ? x0 : O[p == 0 ? 0 : p - 1];
            a = lse2f(E[p], om1);
            cc = lse3f(O[p], E[p], skip[p] ? om1 : NEGINF);
          } else {
            // beta: E_g <- lse(E_g, O_g) + e_blank ; O_g <- lse(O_g, E_{g+1}, skip ? O_{g+1} : -inf) + e_label
            const float ep1 = (p == P - 1) ? x0 : E[p == P - 1 ? p : p + 1];
            const float op1 = (p == P - 1) ? x1 : O[p == P - 1 ? p : p + 1];
            a = lse2f(E[p], O[p]);
            cc = lse3f(O[p], ep1, skip[p] ? op1 : NEGINF);
          }
          nE[p] = (vE[p] && a != NEGINF) ? a + eb : NEGINF;
          nO[p] = (vO[p] && cc != NEGINF) ? cc + el[p] : NEGINF;
        }
        if (step == 0) {  // (wave-uniform) alpha_0: states 0, 1 ; beta_{T-1}: states S-1, S-2
#pragma unroll
          for (int p = 0; p < P; ++p) {
            const int g = lane * P + p;
            nE[p] = (g == (BETA ? U : 0)) ? eb : NEGINF;
            nO[p] = (vO[p] && g == (BETA ? U - 1 : 0)) ? el[p] : NEGINF;
          }
        }
        float* wrow = ws + (long long)t * Smax + 2 * lane * P;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          E[p] = nE[p]; O[p] = nO[p];
          if (vE[p]) wrow[2 * p] = nE[p] * CTC_LN2;      // (-inf stays -inf)
          if (vO[p]) wrow[2 * p + 1] = nO[p] * CTC_LN2;
        }
      }
    }
    __syncthreads();   // chunk c + 1 is in LDS; the walker is done with chunk c's half
  }
  if (!BETA && !loader) {
    // log-likelihood from the last alpha row: states S-1 = E_U and S-2 = O_{U-1}
    float eU = NEGINF, oU = NEGINF;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int g = lane * P + p;
      if (g == U) eU = E[p];
      if (g == U - 1) oU = O[p];
    }
    const float l1 = __shfl(eU, U / P, 64) * CTC_LN2;
    const float l2 = (U >= 1) ? __shfl(oU, (U - 1) / P, 64) * CTC_LN2 : NEGINF;
    if (lane == 0) {
      const float m = fmaxf(l1, l2);
      const float ll = (m == NEGINF) ? NEGINF : m + logf(expf(l1 - m) + expf(l2 - m));
      float out = -ll;
      if (out == INFINITY && zero_infinity) out = 0.f;
      *nll_out = out;
    }
  }
}

// grid (B, 2), two waves per workgroup (walker + emission loader): blockIdx.y = 0 walks alpha, 1 walks beta
template <int P>
__global__ __launch_bounds__(128) void ctc_wave_kernel(const float* __restrict__ logp, const long long* __restrict__ targets,
                                                       const long long* __restrict__ in_len, const long long* __restrict__ tgt_len,
                                                       float* __restrict__ alpha_ws, float* __restrict__ beta_ws,
                                                       float* __restrict__ nll_out, int Tmax, int C, int Umax, int Smax, int blank,
                                                       int zero_infinity) {
  const int b = blockIdx.x;
  const int T = (int)min((long long)Tmax, in_len[b]);
  const int U = (int)min((long long)Umax, tgt_len[b]);
  if (T <= 0) {  // empty input: feasible only for an empty target
    if (threadIdx.x == 0 && blockIdx.y == 0) nll_out[b] = (U == 0) ? 0.f : (zero_infinity ? 0.f : INFINITY);
    return;
  }
  __shared__ float s_e[2][CtcTc<P>::v * (64 * P + 1)];   // double-buffered emission chunks
  const float* lp = logp + (long long)b * Tmax * C;
  const long long* tg = targets + (long long)b * Umax;
  if (blockIdx.y == 0)
    ctc_wave_walk<P, false>(lp, tg, alpha_ws + (long long)b * Tmax * Smax, nll_out + b, T, U, C, Smax, blank, zero_infinity, s_e);
  else
    ctc_wave_walk<P, true>(lp, tg, beta_ws + (long long)b * Tmax * Smax, nullptr, T, U, C, Smax, blank, zero_infinity, s_e);
}

// MI355X_CTC_WAVE (mi355x_ctc_config): 1 (default) = the wave-resident lattice kernel for S <= 1024 states, 0 = the LDS / barrier form
static int g_ctc_wave = -1;
extern "C" int mi355x_ctc_config(int wave) {
  const int prev = g_ctc_wave;
  g_ctc_wave = wave;
  return prev;
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
  if (g_ctc_wave < 0) {
    const char* e = getenv("MI355X_CTC_WAVE");
    g_ctc_wave = (e && e[0] == '0') ? 0 : 1;
  }
#define CTC_WAVE_LAUNCH(PP)                                                                                                        \
  MI_LAUNCH(ctc_wave_kernel<PP>, dim3(B, 2), dim3(128), 0, (hipStream_t)stream, (const float*)logp, (const long long*)targets,        \
            (const long long*)in_len, (const long long*)tgt_len, (float*)alpha_ws, (float*)beta_ws, (float*)nll, Tmax, C, Umax,    \
            Smax, blank, zero_infinity)
  if (g_ctc_wave && Smax <= 128) { CTC_WAVE_LAUNCH(1); }
  else if (g_ctc_wave && Smax <= 256) { CTC_WAVE_LAUNCH(2); }
  else if (g_ctc_wave && Smax <= 512) { CTC_WAVE_LAUNCH(4); }
  else if (g_ctc_wave && Smax <= 1024) { CTC_WAVE_LAUNCH(8); }
  else
  MI_LAUNCH(ctc_kernel, dim3(B), dim3(256), shm, (hipStream_t)stream, (const float*)logp, (const long long*)targets,
                     (const long long*)in_len, (const long long*)tgt_len, (float*)alpha_ws, (float*)beta_ws, (float*)nll, Tmax,
                     C, Umax, Smax, blank, zero_infinity);
#undef CTC_WAVE_LAUNCH
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

