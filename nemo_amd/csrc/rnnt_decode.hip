// Greedy RNN-Transducer decoding on the device (FastConformer-Transducer, BASELINE.json configs[3]): the whole auto-regressive
// search of a batch in ONE launch, no host round trip per frame or per symbol.
//
// Replaces on the reference path: GreedyBatchedRNNTInfer (nemo/collections/asr/parts/submodules/rnnt_greedy_decoding.py:529-990;
// `_greedy_decode_blank_as_pad_loop_frames` :804-990 and the label-looping computer produce the same hypotheses), i.e. per
// utterance the textbook greedy search:
//     state = 0, last = blank (the start-of-sequence input is the zero embedding: blank_as_pad, rnnt.py:880-883)
//     for t < T_b:  repeat up to max_symbols times:
//         g = pred(emb[last], state)  ->  logits = out(relu(enc_proj[t] + pred_proj(g)))        (rnnt.py:1640-1720, joint_net = ReLU,
//         k = argmax logits;  blank -> next frame;  else emit (k, t), state <- state', last <- k          Dropout(eval: off), Linear)
// The reference runs it as a Python loop over frames with a device->host sync per inner iteration (`blank_mask.all()`, :931) and
// B x (prediction step + joint step) launches each; here a workgroup owns an utterance, keeps the LSTM state, the prediction
// projection and the logits in LDS, and streams the weights (fp32 masters, or the bf16 GEMM images) from L2 as matrix-vector
// products -- an HBM / L2-bandwidth-bound loop by nature (no reuse across the 1-row "batch" of a workgroup: not MFMA work).
// Outputs are integer token ids and frame indices: bit-exact against the oracle restatement (oracle/transducer_ref.py
// greedy_decode) and the reference-run fixture (tests/golden/ref_rnnt_greedy.json).
#include "common.h"
#include "mi355x_asr.h"

#define RD_THREADS 512
#define RD_WAVES 8

template <typename WT> __device__ __forceinline__ void rd_ld4(const WT* p, float (&v)[4]);
template <> __device__ __forceinline__ void rd_ld4<float>(const float* p, float (&v)[4]) { ld4<float>(p, v); }
template <> __device__ __forceinline__ void rd_ld4<bf16_t>(const bf16_t* p, float (&v)[4]) { ld4<bf16_t>(p, v); }

// out[r] (+)= sum_k W[r * ldw + k] * v[k] for r in [0, R): one wave per row (rows dealt round-robin to the 8 waves), lanes stride
// over k in 4-element vectors, fp32 accumulation, wave_sum.  `v` lives in LDS; K % 4 == 0.
template <typename WT, bool ACC>
__device__ __forceinline__ void rd_gemv(const WT* __restrict__ W, long long ldw, const float* v, int R, int K, const float* bias,
                                        float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < R; r += RD_WAVES) {
    const WT* w = W + (long long)r * ldw;
    float a = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
      float x[4];
      rd_ld4<WT>(w + k, x);
      const float4 y = *reinterpret_cast<const float4*>(v + k);
      a += x[0] * y.x + x[1] * y.y + x[2] * y.z + x[3] * y.w;
    }
    a = wave_sum(a);
    if (lane == 0) out[r] = (ACC ? out[r] : 0.f) + a + (bias ? bias[r] : 0.f);
  }
}

struct RnntDecP {
  const void* f; int f_dt; long long ldf;   // encoder projection [B, T, J] (rows of pitch ldf)
  const long long* enc_len;
  const float* emb;                          // [V1, H] fp32 (row `blank` is the zero padding row)
  const void *w_ih, *w_hh, *w_pred, *w_out;  // [4H, H], [4H, H], [J, H], [V1, J] (dtype w_dt, row pitches below)
  long long ld_ih, ld_hh, ld_pred, ld_out;
  const float *b_ih, *b_hh, *b_pred, *b_out;
  int* tokens; int* times; int* out_len; float* score;
  float* h_out; float* c_out;                // optional final state [B, H]
  int B, T, J, H, V1, blank, max_symbols, max_out;
};

template <typename WT>
__global__ __launch_bounds__(RD_THREADS) void rnnt_greedy_kernel(RnntDecP p) {
  extern __shared__ __attribute__((aligned(16))) float rd_smem[];
  const int H = p.H, J = p.J, V1 = p.V1;
  float* x = rd_smem;            // [H] embedding of the last emitted label
  float* h = x + H;              // committed state
  float* c = h + H;
  float* hn = c + H;             // state after consuming `last` (committed when the next label is emitted)
  float* cn = hn + H;
  float* z = cn + H;             // [4H] gate pre-activations
  float* gp = z + 4 * H;         // [J] prediction projection of hn
  float* a = gp + J;             // [J] relu(f_t + gp)
  float* lg = a + J;             // [V1] logits
  __shared__ float red_v[RD_WAVES];
  __shared__ int red_i[RD_WAVES];
  __shared__ int s_k;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int len = (int)min((long long)p.T, max(0LL, p.enc_len ? p.enc_len[b] : (long long)p.T));
  for (int i = tid; i < H; i += RD_THREADS) { h[i] = 0.f; c[i] = 0.f; x[i] = 0.f; }
  __syncthreads();

  // prediction step on (x, h, c): hn, cn, gp.  torch gate order i, f, g, o (common/parts/rnn.py:151-230 -> torch.nn.LSTM)
  auto pred_step = [&](bool x_zero) {
    rd_gemv<WT, false>((const WT*)p.w_hh, p.ld_hh, h, 4 * H, H, p.b_hh, z);
    __syncthreads();
    if (!x_zero) rd_gemv<WT, true>((const WT*)p.w_ih, p.ld_ih, x, 4 * H, H, p.b_ih, z);
    else for (int i = tid; i < 4 * H; i += RD_THREADS) z[i] += p.b_ih[i];
    __syncthreads();
    for (int j = tid; j < H; j += RD_THREADS) {
      const float ig = 1.f / (1.f + expf(-z[j])), fg = 1.f / (1.f + expf(-z[H + j]));
      const float gg = tanhf(z[2 * H + j]), og = 1.f / (1.f + expf(-z[3 * H + j]));
      const float cc = fg * c[j] + ig * gg;
      cn[j] = cc;
      hn[j] = og * tanhf(cc);
    }
    __syncthreads();
    rd_gemv<WT, false>((const WT*)p.w_pred, p.ld_pred, hn, J, H, p.b_pred, gp);
    __syncthreads();
  };
  pred_step(true);

  int n = 0;
  float score = 0.f;
  for (int t = 0; t < len; ++t) {
    const char* frow = (const char*)p.f + ((long long)b * p.T + t) * p.ldf * (p.f_dt == MI_DT_F32 ? 4 : 2);
    // max_symbols <= 0 (the reference's `max_symbols_per_step=None`: unbounded inner loop, rnnt_greedy_decoding.py:620-700) is
    // bounded HERE by the output budget: a model that never emits blank would otherwise keep this workgroup -- and the GPU --
    // busy for ever.  Once max_out labels are out the search stops (out_len == max_out tells the caller it was cut short).
    for (int sym = 0; p.max_symbols > 0 ? sym < p.max_symbols : n < p.max_out; ++sym) {
      for (int j = tid; j < J; j += RD_THREADS) {
        const float fv = p.f_dt == MI_DT_F32 ? ((const float*)frow)[j] : bf2f(((const bf16_t*)frow)[j]);
        a[j] = fmaxf(fv + gp[j], 0.f);
      }
      __syncthreads();
      rd_gemv<WT, false>((const WT*)p.w_out, p.ld_out, a, V1, J, p.b_out, lg);
      __syncthreads();
      // arg-max (lowest index among equal maxima, like torch.max) and log-sum-exp (score = sum of the emitted labels' log-probs)
      float bv = -INFINITY; int bi = 0x7fffffff;
      for (int v = tid; v < V1; v += RD_THREADS) { const float q = lg[v]; if (q > bv) { bv = q; bi = v; } }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      if (lane == 0) { red_v[wave] = bv; red_i[wave] = bi; }
      __syncthreads();
      float mv = red_v[0]; int mi = red_i[0];
#pragma unroll
      for (int w = 1; w < RD_WAVES; ++w) if (red_v[w] > mv || (red_v[w] == mv && red_i[w] < mi)) { mv = red_v[w]; mi = red_i[w]; }
      float se = 0.f;
      for (int v = tid; v < V1; v += RD_THREADS) se += expf(lg[v] - mv);
      __syncthreads();   // (red_v is re-used by block_sum)
      se = block_sum(se, red_v);
      if (tid == 0) s_k = mi;
      __syncthreads();
      const int k = s_k;
      if (k == p.blank) break;
      if (n < p.max_out) {
        if (tid == 0) { p.tokens[(long long)b * p.max_out + n] = k; if (p.times) p.times[(long long)b * p.max_out + n] = t; }
      }
      ++n;
      // Score semantics: the sum of the emitted labels' LOG-PROBABILITIES (log-softmax of the joint's logits) -- what the
      // reference's search computes on CPU tensors.  On CUDA tensors its `_joint_step(log_normalize=None)` skips the
      // log-softmax and sums raw maximum logits instead (rnnt_greedy_decoding.py:257-259, 965): scores then differ by the
      // summed log-partition terms; the token ids, time stamps and lengths are the same either way.
      score += -logf(se);   // log-prob of the arg-max label = mv - (mv + log se)
      // commit the state that consumed the previous label, consume the new one
      for (int i = tid; i < H; i += RD_THREADS) { h[i] = hn[i]; c[i] = cn[i]; x[i] = p.emb[(long long)k * H + i]; }
      __syncthreads();
      pred_step(false);
    }
  }
  if (tid == 0) { p.out_len[b] = n < p.max_out ? n : p.max_out; if (p.score) p.score[b] = score; }
  for (int i = tid + n; i < p.max_out; i += RD_THREADS) {   // pad (also when nothing was emitted)
    p.tokens[(long long)b * p.max_out + i] = -1;
    if (p.times) p.times[(long long)b * p.max_out + i] = -1;
  }
  if (p.h_out) for (int i = tid; i < H; i += RD_THREADS) { p.h_out[(long long)b * H + i] = h[i]; p.c_out[(long long)b * H + i] = c[i]; }
}

extern "C" int mi355x_rnnt_greedy_decode(const void* enc_proj, int f_dtype, long long ldf, const void* enc_len, const void* emb,
                                         const void* w_ih, long long ld_ih, const void* w_hh, long long ld_hh, const void* b_ih,
                                         const void* b_hh, const void* w_pred, long long ld_pred, const void* b_pred,
                                         const void* w_out, long long ld_out, const void* b_out, int w_dtype, int B, int T, int J,
                                         int H, int V1, int blank, int max_symbols, void* tokens, void* times, void* out_len,
                                         void* score, int max_out, void* h_out, void* c_out, void* stream) {
  mi_clear_errors();
  if (!enc_proj || !emb || !w_ih || !w_hh || !b_ih || !b_hh || !w_pred || !w_out || !tokens || !out_len) return MI_ERR_ARG;
  if (B <= 0 || T <= 0 || J <= 0 || H <= 0 || V1 <= 1 || max_out <= 0 || blank < 0 || blank >= V1) return MI_ERR_ARG;
  if ((H & 3) || (J & 3) || (ld_ih & 3) || (ld_hh & 3) || (ld_pred & 3) || (ld_out & 3) || (!h_out != !c_out)) return MI_ERR_ARG;
  if ((f_dtype != MI_DT_F32 && f_dtype != MI_DT_BF16) || (w_dtype != MI_DT_F32 && w_dtype != MI_DT_BF16)) return MI_ERR_ARG;
  RnntDecP p;
  p.f = enc_proj; p.f_dt = f_dtype; p.ldf = ldf; p.enc_len = (const long long*)enc_len; p.emb = (const float*)emb;
  p.w_ih = w_ih; p.w_hh = w_hh; p.w_pred = w_pred; p.w_out = w_out;
  p.ld_ih = ld_ih; p.ld_hh = ld_hh; p.ld_pred = ld_pred; p.ld_out = ld_out;
  p.b_ih = (const float*)b_ih; p.b_hh = (const float*)b_hh; p.b_pred = (const float*)b_pred; p.b_out = (const float*)b_out;
  p.tokens = (int*)tokens; p.times = (int*)times; p.out_len = (int*)out_len; p.score = (float*)score;
  p.h_out = (float*)h_out; p.c_out = (float*)c_out;
  p.B = B; p.T = T; p.J = J; p.H = H; p.V1 = V1; p.blank = blank; p.max_symbols = max_symbols; p.max_out = max_out;
  const size_t shm = (size_t)(9 * H + 2 * J + V1 + 4) * sizeof(float);
  if (shm > 160 * 1024 - 256) return MI_ERR_ARG;
  if (w_dtype == MI_DT_F32) {
    hipFuncSetAttribute((const void*)rnnt_greedy_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    MI_LAUNCH(rnnt_greedy_kernel<float>, dim3(B), dim3(RD_THREADS), shm, (hipStream_t)stream, p);
  } else {
    hipFuncSetAttribute((const void*)rnnt_greedy_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    MI_LAUNCH(rnnt_greedy_kernel<bf16_t>, dim3(B), dim3(RD_THREADS), shm, (hipStream_t)stream, p);
  }
  return mi_check_launch();
}
