// RNN-Transducer loss + gradient w.r.t. the joint network's LOGITS (log-softmax fused), SURVEY.md section 8f row 3.
//
//   acts [B, T, U1, V1] f32 logits (U1 = max label length + 1, V1 = vocabulary + blank)
//   kernel 1  rnnt_denom:      per (b,t,u) row: denom = -logsumexp(row); also the two log-probabilities the lattice needs,
//                              lp_blank = denom + row[blank], lp_label = denom + row[labels[b,u]] -- so the recursions never
//                              touch the [B,T,U1,V1] tensor again (one wave per row, 16-byte loads after an alignment peel, wave shuffles)
//   kernel 2  rnnt_lattice:    alpha (blockIdx.y = 0) and beta (= 1) recursions over the (t,u) lattice, one workgroup per
//                              utterance, thread = u, anti-diagonal sweep d = t + u: the left neighbour's value travels
//                              through a double-buffered LDS row (one barrier per diagonal), the emission terms are
//                              prefetched four diagonals ahead into a register ring
//   kernel 3  rnnt_grad:       per (b,t,u) row: the closed-form gradient of -log P(y|x) w.r.t. the logits (softmax
//                              Jacobian folded in), FastEmit term and clamp included; padded cells are written as zeros
//
// Replaces on the reference path (FastConformer-Transducer, cfg 4): the Numba-CUDA kernels
//   nemo/collections/asr/parts/numba/rnnt_loss/utils/cuda_utils/gpu_rnnt_kernel.py:74-407 (alphas :74-183, betas :186-283,
//   grads :286-407), reduce.py (denominator), rnnt_helper.compute_costs_data (:107-116), driven by gpu_rnnt.py:125-231.
#include "common.h"
#include "mi355x_asr.h"

#define RNEG (-INFINITY)

__device__ __forceinline__ float rnnt_lae(float a, float b) {  // log(exp a + exp b), rnnt_helper.log_sum_exp
  if (a == RNEG) return b;
  if (b == RNEG) return a;
  const float m = fmaxf(a, b);
  return m + log1pf(expf(-fabsf(a - b)));
}

// ---- kernel 1: 256 threads = 4 waves = 4 rows
__global__ __launch_bounds__(256) void rnnt_denom_kernel(const float* __restrict__ acts, const long long* __restrict__ labels,
                                                         const long long* __restrict__ xlen, const long long* __restrict__ ylen,
                                                         float* __restrict__ denom, float* __restrict__ lpb,
                                                         float* __restrict__ lpl, long long rows, int T, int U1, int V1,
                                                         int blank, long long ld) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int u = (int)(row % U1);
  const long long bt = row / U1;
  const int t = (int)(bt % T);
  const int b = (int)(bt / T);
  const int Tb = (int)min((long long)T, xlen[b]), Ub = (int)min((long long)(U1 - 1), ylen[b]) + 1;
  if (t >= Tb || u >= Ub) return;  // never read
  const float* x = acts + row * ld;
  // a row starts on a 4-byte boundary only (V1 = 1025 is the common case): up to 3 head scalars bring the walk to a 16-byte
  // boundary, then float4 loads, then up to 3 tail scalars
  const int head = min(V1, (int)((4u - (unsigned)(((unsigned long long)x >> 2) & 3u)) & 3u));
  const int n4 = (V1 - head) >> 2;
  const int tail0 = head + 4 * n4, ntail = V1 - tail0;
  const float4* x4 = reinterpret_cast<const float4*>(x + head);
  float m = RNEG;
  if (lane < head) m = x[lane];
  if (lane < ntail) m = fmaxf(m, x[tail0 + lane]);
  for (int i = lane; i < n4; i += 64) {
    const float4 v = x4[i];
    m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
  }
  m = wave_max(m);
  float s = 0.f;
  if (lane < head) s = __expf(x[lane] - m);
  if (lane < ntail) s += __expf(x[tail0 + lane] - m);
  for (int i = lane; i < n4; i += 64) {  // second pass hits L2 / the TA cache: the row is a few KiB
    const float4 v = x4[i];
    s += (__expf(v.x - m) + __expf(v.y - m)) + (__expf(v.z - m) + __expf(v.w - m));
  }
  s = wave_sum(s);
  if (lane == 0) {
    const float d = -(m + logf(s));
    denom[row] = d;
    lpb[row] = d + x[blank];
    lpl[row] = (u < Ub - 1) ? d + x[labels[(long long)b * (U1 - 1) + u]] : RNEG;
  }
}

// ---- kernel 2: grid (B, 2), blockDim = U1 rounded up to a wave.  LDS: 2 rows of blockDim floats.
__global__ void rnnt_lattice_kernel(const float* __restrict__ lpb, const float* __restrict__ lpl,
                                    const long long* __restrict__ xlen, const long long* __restrict__ ylen,
                                    float* __restrict__ alphas, float* __restrict__ betas, float* __restrict__ ll, int B, int T,
                                    int U1) {
  extern __shared__ float nb[];  // [2][blockDim.x]
  const int b = blockIdx.x;
  const bool is_beta = blockIdx.y == 1;
  const int u = threadIdx.x;
  const int Tb = (int)min((long long)T, xlen[b]), Ub = (int)min((long long)(U1 - 1), ylen[b]) + 1;
  const long long base = (long long)b * T * U1;
  if (Tb <= 0) {
    if (u == 0) ll[(is_beta ? B : 0) + b] = RNEG;
    return;
  }
  const float* pb = lpb + base;
  const float* pl = lpl + base;
  float* out = (is_beta ? betas : alphas) + base;
  const int nd = Tb + Ub - 1;
  // lattice coordinates of this thread on diagonal d:   alpha: (t, u) = (d - u, u)
  //                                                     beta : mirrored, (t, u') with u' = Ub-1-u, t = Tb-1-(d-u)
  const bool active_u = u < Ub;
  const int uu = is_beta ? (Ub - 1 - u) : u;
  // emission terms of step d for this thread (both are needed by the cell computed at diagonal d):
  //   alpha cell (t,u): blank from (t-1,u), label from (t,u-1);   beta cell (t,u'): blank at (t,u'), label at (t,u')
  auto fetch = [&](int d, float& eb, float& el) {
    eb = RNEG; el = RNEG;
    const int k = d - u;
    if (!active_u || k < 0 || k >= Tb) return;
    if (!is_beta) {
      const int t = k;
      if (t > 0) eb = pb[(long long)(t - 1) * U1 + uu];
      if (uu > 0) el = pl[(long long)t * U1 + uu - 1];
    } else {
      const int t = Tb - 1 - k;
      eb = pb[(long long)t * U1 + uu];
      if (uu < Ub - 1) el = pl[(long long)t * U1 + uu];
    }
  };
  float rb[4], rl[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) fetch(i, rb[i], rl[i]);
  float mine = RNEG;  // this thread's cell on the previous diagonal = (t-1,u) for alpha, (t+1,u') for beta
  nb[u] = RNEG; nb[blockDim.x + u] = RNEG;
  __syncthreads();
  for (int d0 = 0; d0 < nd; d0 += 4) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int d = d0 + r;
      if (d < nd) {  // uniform
        const float eb = rb[r], el = rl[r];
        fetch(d + 4, rb[r], rl[r]);
        const int k = d - u;
        float v = RNEG;
        if (active_u && k >= 0 && k < Tb) {
          const float left = (u > 0) ? nb[((d + 1) & 1) * blockDim.x + u - 1] : RNEG;  // neighbour's cell of diagonal d-1
          if (!is_beta) {
            if (k == 0 && u == 0) v = 0.f;
            else v = rnnt_lae(mine + eb, left + el);  // no_emit from (t-1,u), emit from (t,u-1)
            out[(long long)k * U1 + uu] = v;
          } else {
            const int t = Tb - 1 - k;
            if (k == 0 && u == 0) v = eb;  // betas[T-1,U-1] = log_probs[T-1,U-1,blank]
            else v = rnnt_lae(mine + eb, left + el);  // no_emit to (t+1,u'), emit to (t,u'+1)
            out[(long long)t * U1 + uu] = v;
          }
          mine = v;
        }
        nb[(d & 1) * blockDim.x + u] = v;
        __syncthreads();
      }
    }
  }
  // log-likelihoods: forward = alpha[T-1,U-1] + lp_blank[T-1,U-1] (thread u = Ub-1 holds it), backward = beta[0,0]
  if (active_u && u == Ub - 1) {
    if (!is_beta) ll[b] = mine + pb[(long long)(Tb - 1) * U1 + (Ub - 1)];
    else ll[B + b] = mine;
  }
}

// ---- kernel 3: one wave per (b,t,u) row, 4 rows per workgroup.  TG = float: the dense [rows, V1] gradient of the loss module's
// contract (ldg = V1).  TG = bf16: the gradient as the K-contiguous operand of the joint's backward GEMMs, row pitch ldg
// (a multiple of 8, columns [V1, ldg) zero) -- no f32 gradient tensor and no cast pass in the fused joint + loss path.
__device__ __forceinline__ void rnnt_store4(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
__device__ __forceinline__ void rnnt_store4(bf16_t* p, float a, float b, float c, float d) {
  const float v[4] = {a, b, c, d};
  st4(p, v);
}
__device__ __forceinline__ void rnnt_store1(float* p, float a) { *p = a; }
__device__ __forceinline__ void rnnt_store1(bf16_t* p, float a) { st(p, a); }
template <typename TG>
__global__ __launch_bounds__(256) void rnnt_grad_kernel(const float* __restrict__ acts, const long long* __restrict__ labels,
                                                        const long long* __restrict__ xlen, const long long* __restrict__ ylen,
                                                        const float* __restrict__ denom, const float* __restrict__ alphas,
                                                        const float* __restrict__ betas, const float* __restrict__ ll,
                                                        TG* __restrict__ grads, long long rows, int T, int U1, int V1,
                                                        int blank, float fastemit_lambda, float clamp, float scale, int same_align,
                                                        long long ld, long long ldg) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int u = (int)(row % U1);
  const long long bt = row / U1;
  const int t = (int)(bt % T);
  const int b = (int)(bt / T);
  const int Tb = (int)min((long long)T, xlen[b]), Ub = (int)min((long long)(U1 - 1), ylen[b]) + 1;
  TG* g = grads + row * ldg;
  const float* x = acts + row * ld;
  for (int i = V1 + lane; i < ldg; i += 64) rnnt_store1(g + i, 0.f);  // pad columns of a pitched operand row
  // same head / float4 body / tail walk as the denominator kernel; rows of acts and grads share their alignment when the two
  // base pointers do (checked by the host entry: otherwise `same_align` is 0 and the walk is scalar)
  const int head = same_align ? min(V1, (int)((4u - (unsigned)(((unsigned long long)x >> 2) & 3u)) & 3u)) : V1;
  const int n4 = (V1 - head) >> 2;
  const int tail0 = head + 4 * n4, ntail = V1 - tail0;
  if (t >= Tb || u >= Ub) {  // padded cell: zero gradient (the reference starts from a zero-filled tensor)
    for (int i = lane; i < head; i += 64) rnnt_store1(g + i, 0.f);
    if (lane < ntail) rnnt_store1(g + tail0 + lane, 0.f);
    for (int i = lane; i < n4; i += 64) rnnt_store4(g + head + 4 * i, 0.f, 0.f, 0.f, 0.f);
    return;
  }
  const float dn = denom[row], a = alphas[row], be = betas[row], logll = ll[b];
  const int lab = (u < Ub - 1) ? (int)labels[(long long)b * (U1 - 1) + u] : -1;
  const float common = a + be + dn - logll;          // grad = exp(common + x[v]) ...
  const float a_ll = a + dn - logll;                  // alphas + logpk - logll = a_ll + x[v]
  const float beta_next_u = (u < Ub - 1) ? betas[row + 1] : RNEG;
  const float beta_next_t = (t < Tb - 1) ? betas[row + U1] : RNEG;
  // FastEmit: lambda * exp(alpha + y_hat(t,u) + beta(t,u+1) + logpk - logll)
  const bool fe = fastemit_lambda > 0.f && u < Ub - 1;
  const float fe_common = fe ? (a + (dn + x[lab]) + beta_next_u + dn - logll) : 0.f;
  const float l1p = log1pf(fastemit_lambda);
  auto one = [&](int v, float xv) -> float {
    float gr = __expf(common + xv);
    if (fe) gr += fastemit_lambda * __expf(fe_common + xv);
    if (v == blank) {
      if (t == Tb - 1 && u == Ub - 1) gr -= __expf(a_ll + xv);
      if (t < Tb - 1) gr -= __expf(a_ll + xv + beta_next_t);
    }
    if (v == lab) gr -= __expf(l1p + a_ll + xv + beta_next_u);
    if (clamp > 0.f) gr = fminf(fmaxf(gr, -clamp), clamp);
    return gr * scale;
  };
  for (int i = lane; i < head; i += 64) rnnt_store1(g + i, one(i, x[i]));
  if (lane < ntail) rnnt_store1(g + tail0 + lane, one(tail0 + lane, x[tail0 + lane]));
  const float4* x4 = reinterpret_cast<const float4*>(x + head);
  for (int i = lane; i < n4; i += 64) {
    const float4 v = x4[i];
    const int e = head + 4 * i;
    rnnt_store4(g + e, one(e, v.x), one(e + 1, v.y), one(e + 2, v.z), one(e + 3, v.w));
  }
}

__global__ void rnnt_cost_kernel(const float* __restrict__ ll, float* __restrict__ costs, int B, float fastemit_lambda) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) costs[b] = -ll[b] * (1.f + fastemit_lambda);  // rnnt_helper.compute_costs_data
}

static long long rnnt_ws_elems(int B, int T, int U1) {
  return 5LL * B * T * U1 + 2LL * B;  // denom, lp_blank, lp_label, alphas, betas + forward / backward log-likelihoods
}
extern "C" int mi355x_rnnt_workspace_elems(int B, int T, int U1, long long* elems) {
  if (!elems || B <= 0 || T <= 0 || U1 <= 0) return MI_ERR_ARG;
  *elems = rnnt_ws_elems(B, T, U1);
  return 0;
}

static int rnnt_loss_impl(const void* acts, long long ld, const void* labels_, const void* act_lens_, const void* label_lens_, int B,
                          int T, int U1, int V1, int blank, float fastemit_lambda, float clamp, float grad_scale, void* costs_,
                          void* grads_, int grads_dtype, long long ldg, void* workspace_, long long workspace_elems,
                          void* stream) {
  mi_clear_errors();
  const long long* labels = (const long long*)labels_;
  const long long* act_lens = (const long long*)act_lens_;
  const long long* label_lens = (const long long*)label_lens_;
  float* costs = (float*)costs_;
  float* workspace = (float*)workspace_;
  if (!acts || (!labels && U1 > 1) || !act_lens || !label_lens || !costs || !workspace) return MI_ERR_ARG;
  if (B <= 0 || T <= 0 || U1 <= 0 || V1 <= 1 || blank < 0 || blank >= V1 || U1 > 1024) return MI_ERR_ARG;
  if (ld < V1 || (grads_ && ldg < V1)) return MI_ERR_ARG;
  if (grads_ && grads_dtype == MI_DT_BF16 && ((ldg & 7) || ((uintptr_t)grads_ & 15))) return MI_ERR_ARG;
  if (workspace_elems < rnnt_ws_elems(B, T, U1)) return MI_ERR_ARG;
  if (clamp < 0.f || fastemit_lambda < 0.f) return MI_ERR_ARG;
  const long long rows = (long long)B * T * U1;
  if ((rows + 3) / 4 > 0x7fffffffLL) return MI_ERR_ARG;
  float* denom = workspace;
  float* lpb = denom + rows;
  float* lpl = lpb + rows;
  float* alphas = lpl + rows;
  float* betas = alphas + rows;
  float* ll = betas + rows;
  hipStream_t s = (hipStream_t)stream;
  const unsigned nblk = (unsigned)((rows + 3) / 4);
  MI_LAUNCH(rnnt_denom_kernel, dim3(nblk), dim3(256), 0, s, (const float*)acts, labels, act_lens, label_lens, denom,
                     lpb, lpl, rows, T, U1, V1, blank, ld);
  const int threads = ((U1 + 63) / 64) * 64;
  MI_LAUNCH(rnnt_lattice_kernel, dim3(B, 2), dim3(threads), 2 * threads * sizeof(float), s, lpb, lpl, act_lens,
                     label_lens, alphas, betas, ll, B, T, U1);
  if (grads_ && grads_dtype == MI_DT_BF16) {
    // vector walk when every logit row starts on a 16-byte boundary (then head = 0 and the 4-element groups of the bf16 row
    // are 8-byte aligned as well)
    const int aligned = (((unsigned long long)acts & 15ull) == 0ull && (ld & 3) == 0) ? 1 : 0;
    MI_LAUNCH((rnnt_grad_kernel<bf16_t>), dim3(nblk), dim3(256), 0, s, (const float*)acts, labels, act_lens, label_lens,
                       denom, alphas, betas, ll, (bf16_t*)grads_, rows, T, U1, V1, blank, fastemit_lambda, clamp, grad_scale,
                       aligned, ld, ldg);
  } else if (grads_) {
    const int same = ((((unsigned long long)acts ^ (unsigned long long)grads_) & 15ull) == 0ull && ((ld - ldg) & 3) == 0) ? 1 : 0;
    MI_LAUNCH((rnnt_grad_kernel<float>), dim3(nblk), dim3(256), 0, s, (const float*)acts, labels, act_lens, label_lens,
                       denom, alphas, betas, ll, (float*)grads_, rows, T, U1, V1, blank, fastemit_lambda, clamp, grad_scale, same,
                       ld, ldg);
  }
  MI_LAUNCH(rnnt_cost_kernel, dim3((B + 63) / 64), dim3(64), 0, s, ll, costs, B, fastemit_lambda);
  return mi_check_launch();
}

extern "C" int mi355x_rnnt_loss(const void* acts, const void* labels, const void* act_lens, const void* label_lens, int B, int T,
                                int U1, int V1, int blank, float fastemit_lambda, float clamp, float grad_scale, void* costs,
                                void* grads, void* workspace, long long workspace_elems, void* stream) {
  return rnnt_loss_impl(acts, V1, labels, act_lens, label_lens, B, T, U1, V1, blank, fastemit_lambda, clamp, grad_scale, costs,
                        grads, MI_DT_F32, V1, workspace, workspace_elems, stream);
}

extern "C" int mi355x_rnnt_loss_ex(const void* acts, long long ld_acts, const void* labels, const void* act_lens,
                                   const void* label_lens, int B, int T, int U1, int V1, int blank, float fastemit_lambda,
                                   float clamp, float grad_scale, void* costs, void* grads, int grads_dtype, long long ld_grads,
                                   void* workspace, long long workspace_elems, void* stream) {
  if (grads_dtype != MI_DT_F32 && grads_dtype != MI_DT_BF16) return MI_ERR_ARG;
  return rnnt_loss_impl(acts, ld_acts, labels, act_lens, label_lens, B, T, U1, V1, blank, fastemit_lambda, clamp, grad_scale, costs,
                        grads, grads_dtype, ld_grads, workspace, workspace_elems, stream);
}
