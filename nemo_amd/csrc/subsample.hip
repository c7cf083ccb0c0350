// Convolution sub-sampling (x4, 'striding'), memory-bound part.  Activation layout is channels-last:
//   mel [B,F,T] f32 (the preprocessor's layout; no transpose is ever materialised)
//   out1 [B,T1,F1,C]   = mask1 * ReLU(Conv2d(1->C, 3x3, s2, p1)(mask0 * mel^T))      direct kernel (K = 9, write-bound)
//   col  [B*T2*F2, 9C] = im2col(out1), k = (kh*3+kw)*C + ci                         feeds the MFMA GEMM for conv2
//   out2 [B,T2,F2,C]   = mask2 * ReLU(col @ W2p^T + b2)                              (GEMM epilogue EPI_RELU_MASK)
// and the matching backward pieces (conv1 weight/bias grads, col2im with the ReLU gate).
//
// Replaces on the reference path: ConvSubsampling.forward / MaskedConvSequential
//   (nemo/collections/asr/parts/submodules/subsampling.py:385-436, 725-759) = F.conv2d x2 + 4 mask multiplies.
#include "common.h"
#include "mi355x_asr.h"

#define DISPATCH_DT(dt, T, ...)                                      \
  if ((dt) == MI_DT_F32) { typedef float T; __VA_ARGS__; }           \
  else { typedef bf16_t T; __VA_ARGS__; }


// ------------------------------------------------------------------------------------------------ conv1 forward
// grid (ceil(F1/8), T1, B); 256 threads over channels
#define C1_TR 4  // output rows (t1) per block
template <typename TO>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ mel, const float* __restrict__ w,
                                                        const float* __restrict__ bias, TO* __restrict__ out,
                                                        const long long* __restrict__ len0, const long long* __restrict__ len1,
                                                        int B, int F, int T, int T1, int F1, int C, int pad) {
  // pad = zero rows / columns in FRONT of the grid: 1 = Conv2d(padding = 1); 2 = CausalConv2D (causal_convs.py:24-72: F.pad(2, 1)
  // on time AND frequency, then no padding); one behind it either way.
  // thread = V consecutive output channels (one 16-byte store per (t1, f1); 2-byte stores run at a fraction of the HBM
  // rate) x every 4th f1; the 9 x V weights stay in registers for the block's C1_TR output rows.
  constexpr int V = VecIO<TO>::V;
  extern __shared__ float patch[];  // [2*C1_TR + 1][F + pad + 1]: mel rows 2*t1_0-pad .., freq shifted by +pad, zero borders
  const int FW = F + pad + 1, NR = 2 * C1_TR + 1;
  const int b = blockIdx.y, t1_0 = blockIdx.x * C1_TR;
  const int tlim = (int)min((long long)T, len0[b]);
  for (int i = threadIdx.x; i < NR * FW; i += 256) {
    const int rr = i / FW, fi = i - rr * FW;
    const int t = 2 * t1_0 - pad + rr, f = fi - pad;
    float v = 0.f;
    if (t >= 0 && t < tlim && f >= 0 && f < F) v = mel[((long long)b * F + f) * T + t];
    patch[i] = v;
  }
  __syncthreads();
  const int CP = C / V;                       // channel chunks
  const int FS = 256 / CP;                    // f1 positions in flight (the launcher enforces CP <= 256)
  const int ck = threadIdx.x % CP, fs = threadIdx.x / CP;
  if (fs >= FS) return;
  const int l1 = (int)min((long long)T1, len1[b]);
  {
    const int c = ck * V;
    float wk[9][V], bs[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      bs[j] = bias[c + j];
#pragma unroll
      for (int k = 0; k < 9; ++k) wk[k][j] = w[(c + j) * 9 + k];
    }
    for (int tr = 0; tr < C1_TR; ++tr) {
      const int t1 = t1_0 + tr;
      if (t1 >= T1) break;
      const bool tvalid = t1 < l1;
      const float* prow = patch + 2 * tr * FW;
      for (int f1 = fs; f1 < F1; f1 += FS) {
        float m[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) m[kh * 3 + kw] = prow[kh * FW + 2 * f1 + kw];
        float o[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
          float a = bs[j];
#pragma unroll
          for (int k = 0; k < 9; ++k) a = fmaf(wk[k][j], m[k], a);
          o[j] = (tvalid && a > 0.f) ? a : 0.f;
        }
        VecIO<TO>::store(out + (((long long)b * T1 + t1) * F1 + f1) * C + c, o);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ conv1 backward (params only)
// dout1 [B,T1,F1,C] (already gated by ReLU and masks) : dw[c,kh,kw] += sum dout1 * x ; db[c] += sum dout1
// grid (ceil(T1/16), B); threads over channels; each block walks 16 t1 x all f1
#define C1_TB 32
template <typename TO>
__global__ __launch_bounds__(256) void conv1_bwd_kernel(const TO* __restrict__ dout, const float* __restrict__ mel,
                                                        const long long* __restrict__ len0, float* __restrict__ dw,
                                                        float* __restrict__ db, float* __restrict__ partial, int B, int F, int T,
                                                        int T1, int F1, int C, int pad) {
  // dout [B,T1,F1,C] is 1.3 GB at the Large shape: the kernel is a pure stream over it.  Lane = V consecutive channels
  // (one 16-byte load per (t1,f1)), wave = every 4th output row; the 3x3 input patch comes from an LDS image of the
  // mel rows this block touches (uniform-address reads).
  constexpr int V = VecIO<TO>::V;
  extern __shared__ float smem[];
  const int TW = 2 * C1_TB + 1, FW = F + pad + 1;
  float* mel_s = smem;                    // [FW][TW]: (f+pad, t - tbase), zero borders / beyond len0
  float* red = smem + FW * TW;            // [4][64*V]
  const int b = blockIdx.y;
  const int t1_0 = blockIdx.x * C1_TB, t1_end = min(T1, t1_0 + C1_TB);
  const int tbase = 2 * t1_0 - pad;
  const int tlim = (int)min((long long)T, len0[b]);
  for (int i = threadIdx.x; i < FW * TW; i += 256) {
    const int fi = i / TW, tt = i - fi * TW;
    const int f = fi - pad, t = tbase + tt;
    float v = 0.f;
    if (f >= 0 && f < F && t >= 0 && t < tlim) v = mel[((long long)b * F + f) * T + t];
    mel_s[i] = v;
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int part = blockIdx.y * gridDim.x + blockIdx.x;
  // C / V channel chunks per wave pass; with fewer than 64 chunks (C = 256 in bf16: 32) the remaining lanes take other f1
  // positions of the row instead of idling (16-byte loads from all 64 lanes), and a butterfly adds their sums at the end
  const int CL = (C / V >= 64 || (64 % (C / V))) ? 64 : C / V;
  const int NF = 64 / CL;
  const int lane_c = lane % CL, fsub = lane / CL;
  for (int c0 = 0; c0 < C; c0 += CL * V) {
    const int c = c0 + lane_c * V;
    float gw[9][V], gb[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      gb[j] = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) gw[k][j] = 0.f;
    }
    if (c < C) {
      for (int t1 = t1_0 + wave; t1 < t1_end; t1 += 4) {
        const float* mrow = mel_s + 2 * (t1 - t1_0);
        const TO* gp = dout + ((long long)b * T1 + t1) * F1 * C + c;
#pragma unroll 4
        for (int f1 = fsub; f1 < F1; f1 += NF) {
          float g[V];
          VecIO<TO>::load(gp + (long long)f1 * C, g);
          float m[9];
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) m[kh * 3 + kw] = mrow[(2 * f1 + kw) * TW + kh];
#pragma unroll
          for (int j = 0; j < V; ++j) {
            gb[j] += g[j];
#pragma unroll
            for (int k = 0; k < 9; ++k) gw[k][j] = fmaf(g[j], m[k], gw[k][j]);
          }
        }
      }
    }
    for (int off = CL; off < 64; off <<= 1) {  // (wave-uniform trip count) sums of the lanes that share a channel chunk
#pragma unroll
      for (int j = 0; j < V; ++j) {
        gb[j] += __shfl_xor(gb[j], off, 64);
#pragma unroll
        for (int k = 0; k < 9; ++k) gw[k][j] += __shfl_xor(gw[k][j], off, 64);
      }
    }
    // cross-wave sum, one tap per round; then one plain store per (tap, channel) into this block's slab
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < V; ++j) red[wave * 64 * V + lane_c * V + j] = k < 9 ? gw[k < 9 ? k : 0][j] : gb[j];
      __syncthreads();
      for (int e = threadIdx.x; e < CL * V; e += 256) {
        const int cc = c0 + e;
        if (cc >= C) continue;
        const float v = (red[e] + red[64 * V + e]) + (red[2 * 64 * V + e] + red[3 * 64 * V + e]);
        if (partial) partial[((long long)part * 10 + k) * C + cc] = v;
        else if (k < 9) atomicAdd(dw + cc * 9 + k, v);
        else atomicAdd(db + cc, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ im2col / col2im (3x3, s2, p1)
// in [B,T1,F1,C] -> col [(b,t2,f2), (kh,kw,ci)]; one thread = one 8-channel (or 4 for f32) vector
template <typename TT>
__global__ __launch_bounds__(256) void im2col_kernel(const TT* __restrict__ in, TT* __restrict__ col, int B, int T1, int F1,
                                                     int T2, int F2, int C, int pad) {
  constexpr int V = sizeof(TT) == 2 ? 8 : 4;
  const int cv = C / V;
  const long long total = (long long)B * T2 * F2 * 9 * cv;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % cv);
    long long r = i / cv;
    const int tap = (int)(r % 9); r /= 9;
    const int f2 = (int)(r % F2); r /= F2;
    const int t2 = (int)(r % T2);
    const int b = (int)(r / T2);
    const int kh = tap / 3, kw = tap - kh * 3;
    const int t1 = 2 * t2 + kh - pad, f1 = 2 * f2 + kw - pad;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (t1 >= 0 && t1 < T1 && f1 >= 0 && f1 < F1)
      v = *reinterpret_cast<const u32x4*>(in + (((long long)b * T1 + t1) * F1 + f1) * C + c * V);
    *reinterpret_cast<u32x4*>(col + ((((long long)b * T2 + t2) * F2 + f2) * 9 + tap) * C + c * V) = v;
  }
}
// din[b,t1,f1,ci] = (act[b,t1,f1,ci] > 0) * sum_{valid taps} dcol[(b,t2,f2), (kh,kw,ci)]
template <typename TT>
__global__ __launch_bounds__(256) void col2im_relu_kernel(const TT* __restrict__ dcol, const TT* __restrict__ act,
                                                          TT* __restrict__ din, int B, int T1, int F1, int T2, int F2, int C,
                                                          int pad) {
  const int cv = C >> 2;
  const long long total = (long long)B * T1 * F1 * cv;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % cv) * 4;
    long long r = i / cv;
    const int f1 = (int)(r % F1); r /= F1;
    const int t1 = (int)(r % T1);
    const int b = (int)(r / T1);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int tn = t1 + pad - kh;
      if (tn < 0 || (tn & 1)) continue;
      const int t2 = tn >> 1;
      if (t2 >= T2) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int fn = f1 + pad - kw;
        if (fn < 0 || (fn & 1)) continue;
        const int f2 = fn >> 1;
        if (f2 >= F2) continue;
        float v[4];
        ld4(dcol + ((((long long)b * T2 + t2) * F2 + f2) * 9 + kh * 3 + kw) * C + c, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += v[j];
      }
    }
    float a[4];
    const long long o = (((long long)b * T1 + t1) * F1 + f1) * C + c;
    ld4(act + o, a);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = a[j] > 0.f ? acc[j] : 0.f;
    st4(din + o, acc);
  }
}

// =================================================================================================
static inline int grid_for(long long n) { long long g = (n + 255) / 256; return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g)); }

static inline int half_len(int n, int pad) { return (n + pad + 1 - 3) / 2 + 1; }  // kernel 3, stride 2, pads (pad, 1)

extern "C" int mi355x_subsample_conv1_fwd_pad(const void* mel, const void* w, const void* bias, void* out, int out_dt,
                                              const void* len0, const void* len1, int B, int F, int T, int C, int pad, void* stream) {
  mi_clear_errors();
  if (!mel || !w || !bias || !out || !len0 || !len1 || B <= 0 || F <= 0 || T <= 0 || C <= 0 || pad < 1 || pad > 2) return MI_ERR_ARG;
  const int T1 = half_len(T, pad), F1 = half_len(F, pad);
  const int V = out_dt == MI_DT_BF16 ? 8 : 4;
  if (C % V || C / V > 256) return MI_ERR_ARG;
  dim3 grid((T1 + C1_TR - 1) / C1_TR, B), block(256);
  const size_t shm = (size_t)(2 * C1_TR + 1) * (F + pad + 1) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_DT(out_dt, TO, MI_LAUNCH((conv1_fwd_kernel<TO>), grid, block, shm, s, (const float*)mel, (const float*)w,
                                             (const float*)bias, (TO*)out, (const long long*)len0, (const long long*)len1, B, F,
                                             T, T1, F1, C, pad));
  return mi_check_launch();
}
extern "C" int mi355x_subsample_conv1_fwd(const void* mel, const void* w, const void* bias, void* out, int out_dt,
                                          const void* len0, const void* len1, int B, int F, int T, int C, void* stream) {
  return mi355x_subsample_conv1_fwd_pad(mel, w, bias, out, out_dt, len0, len1, B, F, T, C, 1, stream);
}
extern "C" int mi355x_subsample_conv1_bwd_pad(const void* dout, int dt, const void* mel, const void* len0, void* dw, void* db, int B,
                                              int F, int T, int C, int pad, void* scratch, long long scratch_elems, void* stream) {
  mi_clear_errors();
  if (!dout || !mel || !len0 || !dw || !db || B <= 0 || F <= 0 || T <= 0 || C <= 0 || C % (dt == MI_DT_BF16 ? 8 : 4) || pad < 1 ||
      pad > 2)
    return MI_ERR_ARG;
  const int T1 = half_len(T, pad), F1 = half_len(F, pad);
  dim3 grid((T1 + C1_TB - 1) / C1_TB, B), block(256);
  const int nparts = grid.x * grid.y;
  if (scratch && scratch_elems < (long long)nparts * 10 * C) return MI_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int V = dt == MI_DT_BF16 ? 8 : 4;
  const size_t shm = ((size_t)(F + pad + 1) * (2 * C1_TB + 1) + 4 * 64 * V) * sizeof(float);
  if (shm > 64 * 1024) return MI_ERR_ARG;
  DISPATCH_DT(dt, TO, MI_LAUNCH((conv1_bwd_kernel<TO>), grid, block, shm, s, (const TO*)dout, (const float*)mel,
                                         (const long long*)len0, (float*)dw, (float*)db, (float*)scratch, B, F, T, T1, F1, C, pad));
  if (scratch)
    MI_LAUNCH(tap_reduce_kernel, dim3((10 * C + 255) / 256, 16), dim3(256), 0, s, (const float*)scratch, nparts, 9, C,
                       (float*)dw, (float*)db);
  return mi_check_launch();
}
extern "C" int mi355x_subsample_conv1_bwd(const void* dout, int dt, const void* mel, const void* len0, void* dw, void* db, int B,
                                          int F, int T, int C, void* scratch, long long scratch_elems, void* stream) {
  return mi355x_subsample_conv1_bwd_pad(dout, dt, mel, len0, dw, db, B, F, T, C, 1, scratch, scratch_elems, stream);
}
extern "C" int mi355x_im2col_3x3s2_pad(const void* in, void* col, int dt, int B, int T1, int F1, int C, int pad, void* stream) {
  mi_clear_errors();
  if (!in || !col || B <= 0 || T1 <= 0 || F1 <= 0 || C <= 0 || (C & 7) || pad < 1 || pad > 2) return MI_ERR_ARG;
  const int T2 = half_len(T1, pad), F2 = half_len(F1, pad);
  const long long total = (long long)B * T2 * F2 * 9 * (C / (dt == MI_DT_BF16 ? 8 : 4));
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_DT(dt, TT, MI_LAUNCH((im2col_kernel<TT>), dim3(grid_for(total)), dim3(256), 0, s, (const TT*)in, (TT*)col, B,
                                         T1, F1, T2, F2, C, pad));
  return mi_check_launch();
}
extern "C" int mi355x_im2col_3x3s2(const void* in, void* col, int dt, int B, int T1, int F1, int C, void* stream) {
  return mi355x_im2col_3x3s2_pad(in, col, dt, B, T1, F1, C, 1, stream);
}
extern "C" int mi355x_col2im_3x3s2_relu_pad(const void* dcol, const void* act, void* din, int dt, int B, int T1, int F1, int C,
                                            int pad, void* stream) {
  mi_clear_errors();
  if (!dcol || !act || !din || B <= 0 || T1 <= 0 || F1 <= 0 || C <= 0 || (C & 3) || pad < 1 || pad > 2) return MI_ERR_ARG;
  const int T2 = half_len(T1, pad), F2 = half_len(F1, pad);
  const long long total = (long long)B * T1 * F1 * (C >> 2);
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_DT(dt, TT, MI_LAUNCH((col2im_relu_kernel<TT>), dim3(grid_for(total)), dim3(256), 0, s, (const TT*)dcol,
                                         (const TT*)act, (TT*)din, B, T1, F1, T2, F2, C, pad));
  return mi_check_launch();
}
extern "C" int mi355x_col2im_3x3s2_relu(const void* dcol, const void* act, void* din, int dt, int B, int T1, int F1, int C,
                                        void* stream) {
  return mi355x_col2im_3x3s2_relu_pad(dcol, act, din, dt, B, T1, F1, C, 1, stream);
}
