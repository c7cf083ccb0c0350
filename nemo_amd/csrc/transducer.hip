// Transducer head of FastConformer-Transducer (BASELINE.json configs[3]): prediction network (embedding + LSTM) and joint
// network glue.  The dense contractions (input / recurrent gate projections, encoder / prediction projections, the
// [B*T*(U+1), J] x [J, V+1] output layer and all their gradients) are mi355x_gemm launches; these are the HBM-bound and
// latency-bound pieces between them.  Replaces on the reference path (nemo/collections/asr/modules/rnnt.py):
//   RNNTDecoder.predict :700-830     torch.nn.Embedding(padding_idx = blank) + zero start-of-sequence frame + torch.nn.LSTM
//                                    (common/parts/rnn.py:151-230, gate order i, f, g, o)
//   RNNTJoint.joint_after_projection :1640-1720   f.unsqueeze(2) + g.unsqueeze(1) -> ReLU -> Dropout -> Linear
// Layouts: prediction network time-major [U+1, B, H] (one step's rows are contiguous: the recurrent GEMM of step t reads
// h[t-1] and writes the gate pre-activations of step t in place over the input projection); joint hidden [B, T, U+1, J].
#include "common.h"
#include "mi355x_asr.h"

#define DISPATCH_DT(dt, T, ...)                                      \
  if ((dt) == MI_DT_F32) { typedef float T; __VA_ARGS__; }           \
  else { typedef bf16_t T; __VA_ARGS__; }

static inline int tgrid(long long n) { long long g = (n + 255) / 256; return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); }

// ------------------------------------------------------------------------------------------------ embedding + SOS frame
// out[0, b, :] = 0 ; out[u + 1, b, :] = emb[targets[b, u]]  (the blank id is the padding row: all zeros by construction)
template <typename TT>
__global__ __launch_bounds__(256) void embed_sos_fwd_kernel(const long long* __restrict__ tgt, const float* __restrict__ emb,
                                                            TT* __restrict__ out, int B, int U, int H, int blank) {
  const int hv = H >> 2;
  const long long nv = (long long)(U + 1) * B * hv;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % hv) * 4;
    const long long r = i / hv;
    const int b = (int)(r % B), u1 = (int)(r / B);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (u1 > 0) {
      const long long id = tgt[(long long)b * U + u1 - 1];
      if (id != blank && id >= 0) ld4(emb + id * H + c, v);
    }
    st4(out + r * H + c, v);
  }
}
// demb[targets[b,u], :] += dX[u + 1, b, :]   (torch.nn.Embedding(padding_idx): the padding row receives no gradient)
template <typename TT>
__global__ __launch_bounds__(256) void embed_sos_bwd_kernel(const long long* __restrict__ tgt, const TT* __restrict__ dx,
                                                            float* __restrict__ demb, int B, int U, int H, int blank) {
  const int hv = H >> 2;
  const long long nv = (long long)U * B * hv;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % hv) * 4;
    const long long r = i / hv;
    const int b = (int)(r % B), u = (int)(r / B);
    const long long id = tgt[(long long)b * U + u];
    if (id == blank || id < 0) continue;
    float g[4];
    ld4(dx + ((long long)(u + 1) * B + b) * H + c, g);
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(demb + id * H + c + j, g[j]);
  }
}

// ------------------------------------------------------------------------------------------------ LSTM cell
__device__ __forceinline__ float tanhf_(float x) { return 2.f * sigmoidf_(2.f * x) - 1.f; }
// z f32 [B, 4H] = x W_ih^T + b_ih + h_prev W_hh^T (gates i | f | g | o); adds b_hh; writes the ACTIVATED gates back over z,
// the new cell state, h in f32 and in the GEMM operand dtype
template <typename TT>
__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(float* __restrict__ z, const float* __restrict__ b_hh,
                                                            const float* __restrict__ c_prev, float* __restrict__ c,
                                                            float* __restrict__ h, TT* __restrict__ h_lp, int B, int H) {
  const long long n = (long long)B * H;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int b = (int)(i / H), e = (int)(i - (long long)b * H);
    float* zr = z + (long long)b * 4 * H;
    const float gi = sigmoidf_(zr[e] + b_hh[e]);
    const float gf = sigmoidf_(zr[H + e] + b_hh[H + e]);
    const float gg = tanhf_(zr[2 * H + e] + b_hh[2 * H + e]);
    const float go = sigmoidf_(zr[3 * H + e] + b_hh[3 * H + e]);
    const float cp = c_prev ? c_prev[i] : 0.f;
    const float cn = gf * cp + gi * gg;
    const float hn = go * tanhf_(cn);
    zr[e] = gi; zr[H + e] = gf; zr[2 * H + e] = gg; zr[3 * H + e] = go;
    c[i] = cn;
    h[i] = hn;
    st<TT>(h_lp + i, hn);
  }
}
// dh f32 [B,H] (gradient w.r.t. h_t: from the layer above plus the recurrent path), dc f32 [B,H] in: d/dc_t from step t+1,
// out: d/dc_{t-1};  act = activated gates of step t, c = c_t, c_prev = c_{t-1} (NULL: zero);  dz [B,4H]: pre-activation grads
template <typename TT>
__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float* __restrict__ dh, float* __restrict__ dc,
                                                            const float* __restrict__ act, const float* __restrict__ c,
                                                            const float* __restrict__ c_prev, TT* __restrict__ dz, int B, int H) {
  const long long n = (long long)B * H;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int b = (int)(i / H), e = (int)(i - (long long)b * H);
    const float* ar = act + (long long)b * 4 * H;
    const float gi = ar[e], gf = ar[H + e], gg = ar[2 * H + e], go = ar[3 * H + e];
    const float tc = tanhf_(c[i]);
    const float dht = dh[i];
    const float dct = dc[i] + dht * go * (1.f - tc * tc);
    const float cp = c_prev ? c_prev[i] : 0.f;
    TT* dr = dz + (long long)b * 4 * H;
    st<TT>(dr + e, dct * gg * gi * (1.f - gi));
    st<TT>(dr + H + e, dct * cp * gf * (1.f - gf));
    st<TT>(dr + 2 * H + e, dct * gi * (1.f - gg * gg));
    st<TT>(dr + 3 * H + e, dht * tc * go * (1.f - go));
    dc[i] = dct * gf;
  }
}

// ------------------------------------------------------------------------------------------------ joint network glue
// h[b,t,u,:] = dropout(relu(f[b,t,:] + g[b,u,:]))    f [B,T,J], g [B,U1,J] -> h [B,T,U1,J]   (8 / 4 channels per lane)
template <typename TT>
__global__ __launch_bounds__(256) void joint_combine_fwd_kernel(const TT* __restrict__ f, const TT* __restrict__ g,
                                                                TT* __restrict__ h, DropCfg drop, int B, int T, int U1, int J) {
  drop_resolve(drop);
  constexpr int V = VecIO<TT>::V;
  const int jv = J / V;
  const long long nv = (long long)B * T * U1 * jv;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % jv) * V;
    const long long r = i / jv;            // (b*T + t)*U1 + u
    const int u = (int)(r % U1);
    const long long bt = r / U1;
    const int b = (int)(bt / T);
    float a[V], e[V], o[V];
    VecIO<TT>::load(f + bt * J + c, a);
    VecIO<TT>::load(g + ((long long)b * U1 + u) * J + c, e);
    float dm[V];
    if constexpr (V == 8) drop_mask8(drop, (uint32_t)(r * J + c), dm);  // J % 8 == 0: the 8 elements are one hash group
    else {
#pragma unroll
      for (int j = 0; j < V; ++j) dm[j] = drop_mask(drop, (uint32_t)(r * J + c + j));
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float s = a[j] + e[j];
      o[j] = s > 0.f ? s * dm[j] : 0.f;
    }
    VecIO<TT>::store(h + r * J + c, o);
  }
}
// dpre = dh * (h > 0) * drop.scale, written in place over dh; df[b,t,:] = sum_u dpre (one thread owns a (b,t,chunk) and walks u).
// (dg[b,u,:] = sum_t dpre is a column sum over t of the [T, U1*J] slab of each b: mi355x_colsum.)
template <typename TT>
__global__ __launch_bounds__(256) void joint_combine_bwd_kernel(TT* __restrict__ dh, const TT* __restrict__ h,
                                                                TT* __restrict__ df, float scale, int B, int T, int U1, int J) {
  constexpr int V = VecIO<TT>::V;
  const int jv = J / V;
  const long long nv = (long long)B * T * jv;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % jv) * V;
    const long long bt = i / jv;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (int u = 0; u < U1; ++u) {
      const long long off = (bt * U1 + u) * J + c;
      float d[V], a[V];
      VecIO<TT>::load(dh + off, d);
      VecIO<TT>::load(h + off, a);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        d[j] = a[j] > 0.f ? d[j] * scale : 0.f;
        acc[j] += d[j];
      }
      VecIO<TT>::store(dh + off, d);
    }
    VecIO<TT>::store(df + bt * J + c, acc);
  }
}

// dst[m, 0:Np] = (n < N) ? alpha * src[m, n] : 0   (f32 -> operand dtype, row pitch padded for 16-byte GEMM operand pieces)
template <typename TT>
__global__ __launch_bounds__(256) void cast_rows_kernel(const float* __restrict__ src, long long ld_in, TT* __restrict__ dst,
                                                        long long ld_out, long long M, int N, int Np, float alpha) {
  const long long n = M * Np;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long m = i / Np;
    const int c = (int)(i - m * Np);
    st<TT>(dst + m * ld_out + c, c < N ? alpha * src[m * ld_in + c] : 0.f);
  }
}

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" int mi355x_embed_sos_fwd(const void* targets, const void* emb, void* out, int dt, int B, int U, int H, int blank,
                                    void* stream) {
  mi_clear_errors();
  if (!emb || !out || (!targets && U > 0) || B <= 0 || U < 0 || H <= 0 || (H & 3)) return MI_ERR_ARG;
  DISPATCH_DT(dt, TT, MI_LAUNCH((embed_sos_fwd_kernel<TT>), dim3(tgrid((long long)(U + 1) * B * (H >> 2))), dim3(256), 0,
                                         (hipStream_t)stream, (const long long*)targets, (const float*)emb, (TT*)out, B, U, H, blank));
  return mi_check_launch();
}
extern "C" int mi355x_embed_sos_bwd(const void* targets, const void* dx, int dt, void* demb, int B, int U, int H, int blank,
                                    void* stream) {
  mi_clear_errors();
  if (!dx || !demb || B <= 0 || U < 0 || H <= 0 || (H & 3)) return MI_ERR_ARG;
  if (U == 0) return MI_OK;
  if (!targets) return MI_ERR_ARG;
  DISPATCH_DT(dt, TT, MI_LAUNCH((embed_sos_bwd_kernel<TT>), dim3(tgrid((long long)U * B * (H >> 2))), dim3(256), 0,
                                         (hipStream_t)stream, (const long long*)targets, (const TT*)dx, (float*)demb, B, U, H, blank));
  return mi_check_launch();
}
extern "C" int mi355x_lstm_cell_fwd(void* z, const void* b_hh, const void* c_prev, void* c, void* h, void* h_lp, int lp_dt, int B,
                                    int H, void* stream) {
  mi_clear_errors();
  if (!z || !b_hh || !c || !h || !h_lp || B <= 0 || H <= 0) return MI_ERR_ARG;
  DISPATCH_DT(lp_dt, TT, MI_LAUNCH((lstm_cell_fwd_kernel<TT>), dim3(tgrid((long long)B * H)), dim3(256), 0,
                                            (hipStream_t)stream, (float*)z, (const float*)b_hh, (const float*)c_prev, (float*)c,
                                            (float*)h, (TT*)h_lp, B, H));
  return mi_check_launch();
}
extern "C" int mi355x_lstm_cell_bwd(const void* dh, void* dc, const void* act, const void* c, const void* c_prev, void* dz,
                                    int dz_dt, int B, int H, void* stream) {
  mi_clear_errors();
  if (!dh || !dc || !act || !c || !dz || B <= 0 || H <= 0) return MI_ERR_ARG;
  DISPATCH_DT(dz_dt, TT, MI_LAUNCH((lstm_cell_bwd_kernel<TT>), dim3(tgrid((long long)B * H)), dim3(256), 0,
                                            (hipStream_t)stream, (const float*)dh, (float*)dc, (const float*)act, (const float*)c,
                                            (const float*)c_prev, (TT*)dz, B, H));
  return mi_check_launch();
}
extern "C" int mi355x_joint_combine_fwd(const void* f, const void* g, void* h, int dt, unsigned drop_key, unsigned drop_threshold,
                                        float drop_scale, int B, int T, int U1, int J, void* stream) {
  mi_clear_errors();
  const int V = dt == MI_DT_BF16 ? 8 : 4;
  if (!f || !g || !h || B <= 0 || T <= 0 || U1 <= 0 || J <= 0 || J % V) return MI_ERR_ARG;
  if ((long long)B * T * U1 * J >= (1LL << 32)) return MI_ERR_ARG;  // the dropout counter is 32 bits: sub-batch the joint
  DropCfg d = mi_drop(drop_key, drop_threshold, drop_scale);
  DISPATCH_DT(dt, TT, MI_LAUNCH((joint_combine_fwd_kernel<TT>), dim3(tgrid((long long)B * T * U1 * (J / V))), dim3(256), 0,
                                         (hipStream_t)stream, (const TT*)f, (const TT*)g, (TT*)h, d, B, T, U1, J));
  return mi_check_launch();
}
extern "C" int mi355x_joint_combine_bwd(void* dh, const void* h, void* df, int dt, float drop_scale, int B, int T, int U1, int J,
                                        void* stream) {
  mi_clear_errors();
  const int V = dt == MI_DT_BF16 ? 8 : 4;
  if (!dh || !h || !df || B <= 0 || T <= 0 || U1 <= 0 || J <= 0 || J % V) return MI_ERR_ARG;
  DISPATCH_DT(dt, TT, MI_LAUNCH((joint_combine_bwd_kernel<TT>), dim3(tgrid((long long)B * T * (J / V))), dim3(256), 0,
                                         (hipStream_t)stream, (TT*)dh, (const TT*)h, (TT*)df, drop_scale, B, T, U1, J));
  return mi_check_launch();
}
extern "C" int mi355x_cast_rows(const void* src, long long ld_in, void* dst, int dst_dt, long long ld_out, long long M, int N,
                                int Np, float alpha, void* stream) {
  mi_clear_errors();
  if (!src || !dst || M <= 0 || N <= 0 || Np < N || ld_in < N || ld_out < Np) return MI_ERR_ARG;
  DISPATCH_DT(dst_dt, TT, MI_LAUNCH((cast_rows_kernel<TT>), dim3(tgrid(M * Np)), dim3(256), 0, (hipStream_t)stream,
                                             (const float*)src, ld_in, (TT*)dst, ld_out, M, N, Np, alpha));
  return mi_check_launch();
}
