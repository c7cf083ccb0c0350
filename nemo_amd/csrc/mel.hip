// Fused log-mel front-end for MI355X: dither -> pre-emphasis (+length mask) -> centre-padded framing -> Hann window ->
// 512-point real FFT (as a 256-point complex radix-4 Stockham FFT in LDS, one wave per frame) -> power -> sparse
// mel filterbank -> log(x + guard), in ONE pass over the audio: each workgroup stages the contiguous audio segment
// of its 32 frames in LDS with coalesced loads (4 B/sample read exactly once from HBM, plus the 352-sample halo),
// and writes [n_mels, 32]-frame output tiles through LDS so every HBM store is a 128-B row segment.
// A second tiny kernel does the per-feature masked normalisation (two-pass mean / unbiased std, wave-shuffle
// reductions) and the pad-value fill.
//
// Replaces on the reference path: FilterbankFeatures.forward + normalize_batch
//   (nemo/collections/asr/parts/preprocessing/features.py:423-502, 59-93): torch.stft (hipFFT) + ~12 elementwise
//   launches, complex [B,257,T] intermediate written/read three times.
// Algorithmic HBM bytes: 4*S read + 4*n_mels*T written per utterance (= 96 KB per audio-second at 80 mels).
#include <stdlib.h>
#include "common.h"
#include "mi355x_asr.h"

#define MEL_FR 32        // frames per workgroup (4 waves x 8 frames)
#define NFFT 512
#define NH 256           // complex FFT size

struct cpx { float x, y; };
__device__ __forceinline__ cpx cmul(cpx a, cpx b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return {a.x - b.x, a.y - b.y}; }

// W512[i] = exp(-2*pi*i*I/512), table holds i in [0,256); i in [256,512) by sign symmetry
__device__ __forceinline__ cpx tw512(const cpx* tab, int i) {
  i &= 511;
  cpx w = tab[i & 255];
  if (i & 256) { w.x = -w.x; w.y = -w.y; }
  return w;
}

__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ audio, const long long* __restrict__ audio_len,
                                                     const float* __restrict__ window, int win, int hop,
                                                     const int* __restrict__ fb_start, const int* __restrict__ fb_len,
                                                     const int* __restrict__ fb_off, const float* __restrict__ fb_w, int n_mels,
                                                     float preemph, float dither, uint32_t seed, float log_guard,
                                                     float* __restrict__ out, int B, int S, int T) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int seg = (MEL_FR - 1) * hop + NFFT;
  float* s_audio = smem;                                  // [seg]
  float* s_win = s_audio + ((seg + 3) & ~3);              // [512] window zero-padded & centred
  cpx* s_tw = (cpx*)(s_win + NFFT);                       // [256]
  cpx* s_fft = s_tw + NH;                                 // [4 waves][2][256]
  float* s_pw = (float*)(s_fft + 4 * 2 * NH);             // [4][260]
  float* s_out = s_pw + 4 * 260;                          // [n_mels][MEL_FR + 1]

  const int b = blockIdx.y;
  const int f0 = blockIdx.x * MEL_FR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long len = audio_len[b];
  const float* xa = audio + (long long)b * S;

  // ---- stage: pre-emphasised, length-masked, centre-padded audio segment  (sample t = t_base + i)
  const int t_base = f0 * hop - NFFT / 2;
  for (int i = tid; i < seg; i += 256) {
    const int t = t_base + i;
    float v = 0.f;
    if (t >= 0 && t < S && t < len) {
      float cur = xa[t];
      float prev = (t > 0) ? xa[t - 1] : 0.f;
      if (dither > 0.f) {
        cur += dither * hash_normal(seed, (uint32_t)((long long)b * S + t));
        if (t > 0) prev += dither * hash_normal(seed, (uint32_t)((long long)b * S + t - 1));
      }
      v = (t > 0) ? cur - preemph * prev : cur;
    }
    s_audio[i] = v;
  }
  const int woff = (NFFT - win) / 2;
  for (int i = tid; i < NFFT; i += 256) s_win[i] = (i >= woff && i < woff + win) ? window[i - woff] : 0.f;
  {
    float sn, cs;
    sincospif(-(float)tid / 256.f, &sn, &cs);  // exp(-2*pi*i*tid/512)
    s_tw[tid] = {cs, sn};
  }
  __syncthreads();

  cpx* bufA = s_fft + wave * 2 * NH;
  cpx* bufB = bufA + NH;
  float* pw = s_pw + wave * 260;

  for (int fi = 0; fi < MEL_FR / 4; ++fi) {
    const int fl = wave * (MEL_FR / 4) + fi;   // local frame
    const int fbase = fl * hop;                // offset of the frame's first sample in s_audio
    // ---- load z[m] = xw[2m] + i xw[2m+1]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = lane + 64 * r;
      cpx z = {s_audio[fbase + 2 * m] * s_win[2 * m], s_audio[fbase + 2 * m + 1] * s_win[2 * m + 1]};
      bufA[m] = z;
    }
    __syncthreads();
    // ---- 4 radix-4 Stockham stages: p = 1, 4, 16, 64
    cpx* src = bufA; cpx* dst = bufB;
#pragma unroll
    for (int p = 1; p < NH; p *= 4) {
      const int j = lane;
      const int k = j & (p - 1);
      const int tstep = k * (NFFT / (4 * p));
      cpx v0 = src[j];
      cpx v1 = cmul(src[j + 64], tw512(s_tw, tstep));
      cpx v2 = cmul(src[j + 128], tw512(s_tw, 2 * tstep));
      cpx v3 = cmul(src[j + 192], tw512(s_tw, 3 * tstep));
      cpx a0 = cadd(v0, v2), a1 = csub(v0, v2), a2 = cadd(v1, v3), d = csub(v1, v3);
      cpx a3 = {d.y, -d.x};  // (v1 - v3) * (-i)
      const int j0 = ((j - k) << 2) + k;
      dst[j0] = cadd(a0, a2);
      dst[j0 + p] = cadd(a1, a3);
      dst[j0 + 2 * p] = csub(a0, a2);
      dst[j0 + 3 * p] = csub(a1, a3);
      __syncthreads();
      cpx* t = src; src = dst; dst = t;
    }
    // 4 stages => result back in bufA (src)
    // ---- real-FFT post-processing -> power spectrum (257 bins)
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const int k = lane + 64 * r;
      if (k <= NH) {
        cpx zk = src[k & (NH - 1)];
        cpx zc = src[(NH - k) & (NH - 1)];
        zc.y = -zc.y;
        cpx ze = {0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y)};
        cpx df = {0.5f * (zk.x - zc.x), 0.5f * (zk.y - zc.y)};
        cpx zo = {df.y, -df.x};  // df / i
        cpx w = (k == NH) ? cpx{-1.f, 0.f} : s_tw[k];
        cpx xk = cadd(ze, cmul(w, zo));
        pw[k] = xk.x * xk.x + xk.y * xk.y;
      }
    }
    __syncthreads();
    // ---- sparse mel filterbank + log
    for (int m = lane; m < n_mels; m += 64) {
      const int st0 = fb_start[m], n = fb_len[m];
      const float* wv = fb_w + fb_off[m];
      float acc = 0.f;
      for (int i = 0; i < n; ++i) acc = fmaf(wv[i], pw[st0 + i], acc);
      s_out[m * (MEL_FR + 1) + fl] = logf(acc + log_guard);
    }
    __syncthreads();
  }
  // ---- coalesced store of the [n_mels][32] tile
  for (int i = tid; i < n_mels * MEL_FR; i += 256) {
    const int m = i / MEL_FR, fl = i - m * MEL_FR;
    const int f = f0 + fl;
    if (f < T) out[((long long)b * n_mels + m) * T + f] = s_out[m * (MEL_FR + 1) + fl];
  }
}

// The same front-end with the work of a frame kept inside its wave (round 4).  The kernel above synchronises the whole workgroup
// seven times per frame although everything between the staging and the final store is wave-private (each wave owns its FFT
// buffer, its power spectrum and its columns of the output tile), keeps two FFT buffers per wave and walks a wave's eight frames
// one after the other -- a chain of LDS round trips with nothing to overlap them (216 us per call = 0.31 TB/s, r3 PMC).  Here:
//   * LDS operations of ONE wave execute in program order, so a write -> read hand-over between lanes of the same wave needs no
//     s_barrier, only that the compiler keeps the order and the data has landed: `s_waitcnt lgkmcnt(0)` with a memory clobber;
//   * the radix-4 stages run IN PLACE (all four inputs of every lane are in registers before the wave's first store issues);
//   * a wave carries FPW = 2 frames through the stages together: two independent dependency chains per lane.
//   * the sparse filterbank is staged in LDS once per workgroup (read from global memory inside the tap loop it cost up to 18
//     dependent L2 round trips per (frame, filter) pair) and the staging loop issues the loads of eight samples per thread together.
// Arithmetic and its order per frame are those of the kernel above (equal up to fused-multiply-add contraction: <= 1e-4 on the
// log-mel values, tests/test_kernels_gpu.py).  201 -> 110 us per call at B = 32 x 20 s (133 us with dither): profiles/r4_logmel.md.
#define FB_CAP 1024      // filterbank weights kept in LDS (floats)
#define WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
template <int FPW>
__global__ __launch_bounds__(256) void logmel_wave_kernel(const float* __restrict__ audio, const long long* __restrict__ audio_len,
                                                          const float* __restrict__ window, int win, int hop,
                                                          const int* __restrict__ fb_start, const int* __restrict__ fb_len,
                                                          const int* __restrict__ fb_off, const float* __restrict__ fb_w, int n_mels,
                                                          float preemph, float dither, uint32_t seed, float log_guard,
                                                          float* __restrict__ out, int B, int S, int T) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int seg = (MEL_FR - 1) * hop + NFFT;
  float* s_audio = smem;                                  // [seg]
  float* s_win = s_audio + ((seg + 3) & ~3);              // [512] window zero-padded & centred
  cpx* s_tw = (cpx*)(s_win + NFFT);                       // [256]
  cpx* s_fft = s_tw + NH;                                 // [4 waves][FPW][256]
  float* s_pw = (float*)(s_fft + 4 * FPW * NH);           // [4][FPW][260]
  float* s_out = s_pw + 4 * FPW * 260;                    // [n_mels][MEL_FR + 1]
  float* s_fbw = s_out + n_mels * (MEL_FR + 1);           // [FB_CAP + 4] filter weights (+ 4 zeros: the tap loop reads in fours)
  int* s_fbi = (int*)(s_fbw + FB_CAP + 4);                // [3][n_mels] first bin, taps, offset into the weights

  const int b = blockIdx.y;
  const int f0 = blockIdx.x * MEL_FR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long len = audio_len[b];
  const float* xa = audio + (long long)b * S;

  // the sparse filterbank (500 weights for 80 Slaney triangles over 257 bins) goes to LDS once per workgroup: read from global
  // memory inside the tap loop -- up to 18 dependent L2 round trips per (frame, filter) -- it was three quarters of the kernel
  const int nnz = fb_off[n_mels - 1] + fb_len[n_mels - 1];
  const bool fb_in_lds = nnz <= FB_CAP;  // (a dense user filterbank stays in global memory)
  if (fb_in_lds) {
    for (int i = tid; i < nnz + 4; i += 256) s_fbw[i] = i < nnz ? fb_w[i] : 0.f;
    for (int i = tid; i < n_mels; i += 256) { s_fbi[i] = fb_start[i]; s_fbi[n_mels + i] = fb_len[i]; s_fbi[2 * n_mels + i] = fb_off[i]; }
  }

  // staging, eight samples per thread and trip: the 16 loads of a trip are issued together (the one-sample loop of the round-1
  // kernel paid one HBM round trip per sample and thread: 21 in a row, ~20 us of the ~35 us a workgroup lived)
  const int t_base = f0 * hop - NFFT / 2;
  constexpr int SU = 8;
  for (int i0 = tid; i0 < seg; i0 += 256 * SU) {
    float cur[SU], prev[SU];
    bool ok[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = i0 + 256 * u, t = t_base + i;
      ok[u] = i < seg && t >= 0 && t < S && t < len;
      cur[u] = ok[u] ? xa[t] : 0.f;
      prev[u] = (ok[u] && t > 0) ? xa[t - 1] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = i0 + 256 * u, t = t_base + i;
      if (i < seg) {
        float v = 0.f;
        if (ok[u]) {
          float c = cur[u], q = prev[u];
          if (dither > 0.f) {
            c += dither * hash_normal(seed, (uint32_t)((long long)b * S + t));
            if (t > 0) q += dither * hash_normal(seed, (uint32_t)((long long)b * S + t - 1));
          }
          v = (t > 0) ? c - preemph * q : c;
        }
        s_audio[i] = v;
      }
    }
  }
  const int woff = (NFFT - win) / 2;
  for (int i = tid; i < NFFT; i += 256) s_win[i] = (i >= woff && i < woff + win) ? window[i - woff] : 0.f;
  {
    float sn, cs;
    sincospif(-(float)tid / 256.f, &sn, &cs);  // exp(-2*pi*i*tid/512)
    s_tw[tid] = {cs, sn};
  }
  __syncthreads();

  cpx* buf0 = s_fft + wave * FPW * NH;
  float* pw0 = s_pw + wave * FPW * 260;
  if (lane < 3 * FPW) pw0[(lane / 3) * 260 + 257 + lane % 3] = 0.f;  // the slack slots the four-tap loop may multiply by zero

  for (int fi = 0; fi < MEL_FR / 4 / FPW; ++fi) {
    const int fl0 = wave * (MEL_FR / 4) + fi * FPW;  // first local frame of this pass
#pragma unroll
    for (int f = 0; f < FPW; ++f) {
      const int fbase = (fl0 + f) * hop;
      cpx* buf = buf0 + f * NH;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = lane + 64 * r;
        cpx z = {s_audio[fbase + 2 * m] * s_win[2 * m], s_audio[fbase + 2 * m + 1] * s_win[2 * m + 1]};
        buf[m] = z;
      }
    }
    WAVE_SYNC();
#pragma unroll
    for (int p = 1; p < NH; p *= 4) {
      const int j = lane;
      const int k = j & (p - 1);
      const int tstep = k * (NFFT / (4 * p));
      const cpx w1 = tw512(s_tw, tstep), w2 = tw512(s_tw, 2 * tstep), w3 = tw512(s_tw, 3 * tstep);
      cpx v0[FPW], v1[FPW], v2[FPW], v3[FPW];
#pragma unroll
      for (int f = 0; f < FPW; ++f) {
        const cpx* src = buf0 + f * NH;
        v0[f] = src[j]; v1[f] = src[j + 64]; v2[f] = src[j + 128]; v3[f] = src[j + 192];
      }
      WAVE_SYNC();  // every lane's inputs are in registers before the first in-place store
      const int j0 = ((j - k) << 2) + k;
#pragma unroll
      for (int f = 0; f < FPW; ++f) {
        cpx* dst = buf0 + f * NH;
        const cpx u1 = cmul(v1[f], w1), u2 = cmul(v2[f], w2), u3 = cmul(v3[f], w3);
        cpx a0 = cadd(v0[f], u2), a1 = csub(v0[f], u2), a2 = cadd(u1, u3), d = csub(u1, u3);
        cpx a3 = {d.y, -d.x};  // (v1 - v3) * (-i)
        dst[j0] = cadd(a0, a2);
        dst[j0 + p] = cadd(a1, a3);
        dst[j0 + 2 * p] = csub(a0, a2);
        dst[j0 + 3 * p] = csub(a1, a3);
      }
      WAVE_SYNC();
    }
#pragma unroll
    for (int f = 0; f < FPW; ++f) {
      const cpx* src = buf0 + f * NH;
      float* pw = pw0 + f * 260;
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        const int k = lane + 64 * r;
        if (k <= NH) {
          cpx zk = src[k & (NH - 1)];
          cpx zc = src[(NH - k) & (NH - 1)];
          zc.y = -zc.y;
          cpx ze = {0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y)};
          cpx df = {0.5f * (zk.x - zc.x), 0.5f * (zk.y - zc.y)};
          cpx zo = {df.y, -df.x};  // df / i
          cpx w = (k == NH) ? cpx{-1.f, 0.f} : s_tw[k];
          cpx xk = cadd(ze, cmul(w, zo));
          pw[k] = xk.x * xk.x + xk.y * xk.y;
        }
      }
    }
    WAVE_SYNC();
    // ---- sparse mel filterbank + log: FPW * n_mels (frame, filter) pairs over the 64 lanes
    for (int e = lane; e < FPW * n_mels; e += 64) {
      const int f = e / n_mels, m = e - f * n_mels;
      const float* pw = pw0 + f * 260;
      float acc = 0.f;
      if (fb_in_lds) {
        const int st0 = s_fbi[m], n = s_fbi[n_mels + m];
        const float* wv = s_fbw + s_fbi[2 * n_mels + m];
        // four taps per trip, all eight LDS reads issued before the four multiply-adds; the sum keeps the tap order, a tap past
        // the filter's end multiplies a finite power by a zero weight (pw has 260 slots for 257 bins, the weights 4 zeros of slack)
        for (int i = 0; i < n; i += 4) {
          const float w0 = wv[i], w1 = i + 1 < n ? wv[i + 1] : 0.f, w2 = i + 2 < n ? wv[i + 2] : 0.f, w3 = i + 3 < n ? wv[i + 3] : 0.f;
          const float p0 = pw[st0 + i], p1 = pw[st0 + i + 1], p2 = pw[st0 + i + 2], p3 = pw[st0 + i + 3];
          acc = fmaf(w0, p0, acc); acc = fmaf(w1, p1, acc); acc = fmaf(w2, p2, acc); acc = fmaf(w3, p3, acc);
        }
      } else {
        const int st0 = fb_start[m], n = fb_len[m];
        const float* wv = fb_w + fb_off[m];
        for (int i = 0; i < n; ++i) acc = fmaf(wv[i], pw[st0 + i], acc);
      }
      s_out[m * (MEL_FR + 1) + fl0 + f] = logf(acc + log_guard);
    }
    WAVE_SYNC();  // (the next pass overwrites buf / pw: keep this pass's reads in front of them)
  }
  __syncthreads();
  for (int i = tid; i < n_mels * MEL_FR; i += 256) {
    const int m = i / MEL_FR, fl = i - m * MEL_FR;
    const int f = f0 + fl;
    if (f < T) out[((long long)b * n_mels + m) * T + f] = s_out[m * (MEL_FR + 1) + fl];
  }
}


// Round 6: the same front-end with the FFT IN REGISTERS.  The radix-4 Stockham passes above move every point through the LDS four
// times (8 + 8 b64 accesses per lane and stage, 2- to 4-way bank conflicts in the strided stores of the first two stages) and the
// kernel sat at ~250 LDS wave-instructions per two-frame pass (profiles/r4_logmel.md).  Here a frame is owned by 16 LANES, each
// holding 16 of its 256 complex points (m = l + 16 r): 256 = 16 x 16, so the transform is
//     A[l][q]  = sum_r z[l + 16 r] W16^(r q)            -- a 16-point DFT over a lane's own registers
//     A'[l][q] = A[l][q] * W256^(l q)                   -- 15 twiddles that depend on the lane only: kept in registers
//     Z[q+16s] = sum_l A'[l][q] W16^(l s)               -- ONE 16 x 16 transpose through the LDS, then a DFT in registers again
// and the real-FFT unscramble pairs bin k = q + 16 s with 256 - k = (16 - q) + 16 (15 - s): the partner lane is fixed
// (ds_bpermute, no LDS storage), its registers are read in reverse order.  A wave carries four frames at a time; per frame the
// LDS sees 16 + 16 b64 accesses for the transpose (17-point pitch, frame buffers 128 B apart modulo the bank row: conflict-free
// both ways) instead of 4 x (8 + 8), plus the window/audio reads and the power spectrum for the filterbank.
// The dither is evaluated ONCE per sample (the kernels above compute the previous sample's noise again for the pre-emphasis):
// the dithered signal goes to the LDS, a second pass turns it into the pre-emphasised one.
// Mel stage: lane (frame g, l) owns filters l, 31-l, 32+l, 63-l, ... of its frame -- long and short triangles alternate, so the
// tap loops of a wave end together.  Same DFT, different summation order than the radix-4 kernels: log-mel values agree to ~1e-6.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f cmulv(v2f a, v2f b) {  // (a.x b.x - a.y b.y, a.x b.y + a.y b.x): two packed operations
  const v2f t = a.xx * b;
  return __builtin_elementwise_fma((v2f){-a.y, a.y}, b.yx, t);
}
__device__ __forceinline__ v2f mul_mi(v2f a) { return (v2f){a.y, -a.x}; }  // a * (-i)
// in-place 16-point forward DFT of x[0..15] (output in natural order), 8 radix-4 butterflies + 9 constant twiddles
__device__ __forceinline__ void dft16(v2f* x) {
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
  v2f y[4][4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const v2f a0 = x[b] + x[b + 8], a1 = x[b] - x[b + 8], a2 = x[b + 4] + x[b + 12], a3 = mul_mi(x[b + 4] - x[b + 12]);
    y[b][0] = a0 + a2; y[b][1] = a1 + a3; y[b][2] = a0 - a2; y[b][3] = a1 - a3;
  }
  // y[b][q0] *= W16^(b q0)
  y[1][1] = cmulv(y[1][1], (v2f){C1, -S1}); y[1][2] = cmulv(y[1][2], (v2f){R2, -R2}); y[1][3] = cmulv(y[1][3], (v2f){S1, -C1});
  y[2][1] = cmulv(y[2][1], (v2f){R2, -R2}); y[2][2] = mul_mi(y[2][2]);                y[2][3] = cmulv(y[2][3], (v2f){-R2, -R2});
  y[3][1] = cmulv(y[3][1], (v2f){S1, -C1}); y[3][2] = cmulv(y[3][2], (v2f){-R2, -R2}); y[3][3] = cmulv(y[3][3], (v2f){-C1, S1});
#pragma unroll
  for (int q0 = 0; q0 < 4; ++q0) {
    const v2f a0 = y[0][q0] + y[2][q0], a1 = y[0][q0] - y[2][q0], a2 = y[1][q0] + y[3][q0], a3 = mul_mi(y[1][q0] - y[3][q0]);
    x[q0] = a0 + a2; x[q0 + 4] = a1 + a3; x[q0 + 8] = a0 - a2; x[q0 + 12] = a1 - a3;
  }
}

#define R16_XP 272   // complex slots of a frame's transpose buffer: 16 rows x 17; 2176 B = 128 (mod 256)
__global__ __launch_bounds__(256, 2) void logmel_r16_kernel(const float* __restrict__ audio, const long long* __restrict__ audio_len,
                                                            const float* __restrict__ window, int win, int hop,
                                                            const int* __restrict__ fb_start, const int* __restrict__ fb_len,
                                                            const int* __restrict__ fb_off, const float* __restrict__ fb_w, int n_mels,
                                                            float preemph, float dither, uint32_t seed, float log_guard,
                                                            float* __restrict__ out, int B, int S, int T) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int seg = (MEL_FR - 1) * hop + NFFT;
  float* s_audio = smem;                                  // [seg] pre-emphasised, masked, centre-padded; sample t = t_base + i
  float* s_win = s_audio + ((seg + 3) & ~3);              // [512] window zero-padded & centred
  v2f* s_tw = (v2f*)(s_win + NFFT);                       // [256] exp(-2 pi i k / 512)
  v2f* s_x = s_tw + NH;                                   // [4 waves][4 frames][R16_XP] transpose buffers (later: power spectra)
  float* s_n = (float*)s_x;                               // [seg + 1] dithered signal, sample t = t_base - 1 + i (staging only)
  float* s_out = (float*)(s_x + 4 * 4 * R16_XP);          // [n_mels][MEL_FR + 1]
  float* s_fbw = s_out + n_mels * (MEL_FR + 1);           // [FB_CAP + 4]
  int* s_fbi = (int*)(s_fbw + FB_CAP + 4);                // [3][n_mels]

  const int b = blockIdx.y;
  const int f0 = blockIdx.x * MEL_FR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l = lane & 15, g = lane >> 4;
  const long long len = audio_len[b];
  const float* xa = audio + (long long)b * S;

  // ---- pass 1: dithered signal n[t] (0 outside [0, min(S, len))).  The segment is read as 16-byte vectors aligned on the
  // ABSOLUTE element index (S need not be a multiple of 4), all of a thread's loads in flight before the first use; the
  // filterbank, window and index loads go out behind them, in front of the first wait.
  const int t_base = f0 * hop - NFFT / 2;
  const long long row0 = (long long)b * S;
  const int shift = (int)((row0 + t_base - 1) & 3);      // elements between the aligned start and sample t_base - 1
  const int ta = t_base - 1 - shift;
  const int nvec = (seg + 1 + shift + 3) >> 2;
  const int lim = (int)min((long long)S, len);
  constexpr int SV = 6;                                   // 6 x 256 vectors >= (31 * 512 + 512 + 4) / 4 for every hop <= 512
  float4 av[SV];
#pragma unroll
  for (int u = 0; u < SV; ++u) {
    const int v = tid + 256 * u, t = ta + 4 * v;
    av[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (v < nvec && t + 3 >= 0 && t < lim) {
      if (t >= 0 && t + 3 < S) av[u] = *reinterpret_cast<const float4*>(xa + t);
      else {
        if (t >= 0 && t < S) av[u].x = xa[t];
        if (t + 1 >= 0 && t + 1 < S) av[u].y = xa[t + 1];
        if (t + 2 >= 0 && t + 2 < S) av[u].z = xa[t + 2];
        if (t + 3 >= 0 && t + 3 < S) av[u].w = xa[t + 3];
      }
    }
  }
  const int nnz_x = fb_off[n_mels - 1] + fb_len[n_mels - 1], nnz = (nnz_x + 3) & ~3;  // (weights exist up to nnz_x only)
  const bool fb_in_lds = nnz <= FB_CAP;
  int fb_aligned = 1;
  if (fb_in_lds) {
    for (int i = tid; i < nnz + 4; i += 256) s_fbw[i] = i < nnz_x ? fb_w[i] : 0.f;
    for (int i = tid; i < n_mels; i += 256) {
      const int o = fb_off[i];
      s_fbi[i] = fb_start[i]; s_fbi[n_mels + i] = fb_len[i]; s_fbi[2 * n_mels + i] = o;
      // 16-byte weight reads need 4-aligned offsets and zero fill up to the next filter (sparsify_filterbank lays them out so)
      if ((o & 3) || (i + 1 < n_mels && fb_off[i + 1] < o + ((fb_len[i] + 3) & ~3))) fb_aligned = 0;
    }
  }
#pragma unroll
  for (int u = 0; u < SV; ++u) {
    const int v = tid + 256 * u, t = ta + 4 * v;
    if (v < nvec) {
      const float c[4] = {av[u].x, av[u].y, av[u].z, av[u].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * v - shift + e, te = t + e;
        if (i >= 0 && i < seg + 1) {
          float x = 0.f;
          if (te >= 0 && te < lim) {
            x = c[e];
            if (dither > 0.f) x += dither * hash_normal(seed, (uint32_t)(row0 + te));
          }
          s_n[i] = x;
        }
      }
    }
  }
  const int woff = (NFFT - win) / 2;
  for (int i = tid; i < NFFT; i += 256) s_win[i] = (i >= woff && i < woff + win) ? window[i - woff] : 0.f;
  {
    float sn, cs;
    sincospif(-(float)tid / 256.f, &sn, &cs);
    s_tw[tid] = (v2f){cs, sn};
  }
  const bool fbv = __syncthreads_and(fb_aligned) != 0 && fb_in_lds;  // (also the barrier between the two staging passes)
  if (fbv)  // whatever the caller left between a row's end and the next multiple of four: the vector reads must meet zeros
    for (int m = tid; m < n_mels; m += 256) {
      const int n = s_fbi[n_mels + m], o = s_fbi[2 * n_mels + m];
      for (int k = n; k < ((n + 3) & ~3); ++k) s_fbw[o + k] = 0.f;
    }
  // ---- pass 2: pre-emphasis v[t] = n[t] - preemph * n[t-1] for t in [0, min(S, len)), 0 elsewhere (t = 0: n[-1] = 0)
  for (int i = tid; i < seg; i += 256) {
    const int t = t_base + i;
    const bool ok = t >= 0 && t < S && t < len;
    s_audio[i] = ok ? (t > 0 ? s_n[i + 1] - preemph * s_n[i] : s_n[i + 1]) : 0.f;
  }
  // per-lane constants: W256^(l q) = tw[2 l q] (q = 1..15, index < 512 by sign symmetry) and the unscramble twiddles W512^(l + 16 s)
  v2f twl[16], twp[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = 2 * l * q;
    v2f w = s_tw[i & 255];
    if (i & 256) w = -w;
    twl[q] = w;
    twp[q] = s_tw[l + 16 * q];
  }
  __syncthreads();

  v2f* xb = s_x + (wave * 4 + g) * R16_XP;      // this frame's transpose buffer
  float* pw = (float*)xb;                       // ... and, once the transpose has been read back, its power spectrum [260]
  const int pl = (lane & 48) | ((16 - l) & 15); // the lane that holds bins 256 - k of this lane's bins k

#ifndef R16_ABL
#define R16_ABL 0
#endif
  for (int pass = 0; pass < (R16_ABL == 3 ? 0 : MEL_FR / 16); ++pass) {
    const int fl = wave * (MEL_FR / 4) + pass * 4 + g;  // local frame of this 16-lane group
    const float* fa = s_audio + fl * hop;
    v2f z[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = l + 16 * r;
      const v2f a = *reinterpret_cast<const v2f*>(fa + 2 * m), w = *reinterpret_cast<const v2f*>(s_win + 2 * m);
      z[r] = a * w;
    }
    dft16(z);
#pragma unroll
    for (int q = 1; q < 16; ++q) z[q] = cmulv(z[q], twl[q]);
#pragma unroll
    for (int q = 0; q < 16; ++q) xb[q * 17 + l] = z[q];
    WAVE_SYNC();
#pragma unroll
    for (int j = 0; j < 16; ++j) z[j] = xb[l * 17 + j];   // lane q = l now holds A'[j][q], j = 0..15
    WAVE_SYNC();
    dft16(z);                                             // z[s] = Z[l + 16 s]
    if (R16_ABL == 2) { s_out[l * (MEL_FR + 1) + fl] = z[0].x + z[5].y + z[15].x; continue; }
    // ---- real-FFT unscramble: partner values Z[256 - k]
    v2f zp[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      zp[j].x = __shfl(z[j].x, pl, 64);
      zp[j].y = __shfl(z[j].y, pl, 64);
    }
    const bool l0 = l == 0;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const v2f a = zp[15 - s], c = zp[(16 - s) & 15];
      v2f zc = l0 ? c : a;
      zc.y = -zc.y;
      const v2f zk = z[s];
      const v2f ze = 0.5f * (zk + zc), df = 0.5f * (zk - zc);
      const v2f xk = ze + cmulv(twp[s], mul_mi(df));
      pw[l + 16 * s] = xk.x * xk.x + xk.y * xk.y;
    }
    if (l0) { const float d = z[0].x - z[0].y; pw[256] = d * d; pw[257] = 0.f; pw[258] = 0.f; pw[259] = 0.f; }
    WAVE_SYNC();
    if (R16_ABL == 1) { s_out[l * (MEL_FR + 1) + fl] = pw[l] + pw[l + 100]; continue; }
    // ---- sparse mel filterbank + log
    for (int j = 0; 16 * j < n_mels; ++j) {
      const int m = 16 * j + ((j & 1) ? 15 - l : l);
      if (m >= n_mels) continue;
      float acc = 0.f;
      if (fbv) {
        const int st0 = s_fbi[m], n = s_fbi[n_mels + m];
        const float* wv = s_fbw + s_fbi[2 * n_mels + m];
        const float* pv = pw + st0;
        for (int i = 0; i < n; i += 4) {   // (pw has 260 slots for 257 bins, the weights are zero up to the next multiple of four)
          const float4 w = *reinterpret_cast<const float4*>(wv + i);
          const float p0 = pv[i], p1 = pv[i + 1], p2 = pv[i + 2], p3 = pv[i + 3];
          acc = fmaf(w.x, p0, acc); acc = fmaf(w.y, p1, acc); acc = fmaf(w.z, p2, acc); acc = fmaf(w.w, p3, acc);
        }
      } else if (fb_in_lds) {
        const int st0 = s_fbi[m], n = s_fbi[n_mels + m];
        const float* wv = s_fbw + s_fbi[2 * n_mels + m];
        for (int i = 0; i < n; i += 4) {
          const float w0 = wv[i], w1 = i + 1 < n ? wv[i + 1] : 0.f, w2 = i + 2 < n ? wv[i + 2] : 0.f, w3 = i + 3 < n ? wv[i + 3] : 0.f;
          const float p0 = pw[st0 + i], p1 = pw[st0 + i + 1], p2 = pw[st0 + i + 2], p3 = pw[st0 + i + 3];
          acc = fmaf(w0, p0, acc); acc = fmaf(w1, p1, acc); acc = fmaf(w2, p2, acc); acc = fmaf(w3, p3, acc);
        }
      } else {
        const int st0 = fb_start[m], n = fb_len[m];
        const float* wv = fb_w + fb_off[m];
        for (int i = 0; i < n; ++i) acc = fmaf(wv[i], pw[st0 + i], acc);
      }
      s_out[m * (MEL_FR + 1) + fl] = logf(acc + log_guard);
    }
    WAVE_SYNC();  // (the next pass's transpose overwrites pw)
  }
  __syncthreads();
  for (int i = tid; i < n_mels * MEL_FR; i += 256) {
    const int m = i / MEL_FR, fl = i - m * MEL_FR;
    const int f = f0 + fl;
    if (f < T) out[((long long)b * n_mels + m) * T + f] = s_out[m * (MEL_FR + 1) + fl];
  }
}


// per-feature normalisation over frames t < seq_len (features.py:59-93), pad_value fill beyond; one wave per (b, m) row
template <typename TO>
__global__ __launch_bounds__(256) void feat_norm_kernel(const float* __restrict__ x, const long long* __restrict__ seq_len,
                                                        TO* __restrict__ y, int rows, int n_mels, int T, int normalize,
                                                        float pad_value) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int b = row / n_mels;
  const int n = (int)min((long long)T, seq_len[b]);
  const float* xr = x + (long long)row * T;
  float mu = 0.f, inv = 1.f;
  if (normalize) {
    float s = 0.f;
    for (int t = lane; t < n; t += 64) s += xr[t];
    mu = wave_sum(s) / (float)n;
    float q = 0.f;
    for (int t = lane; t < n; t += 64) { float c = xr[t] - mu; q += c * c; }
    float sd = sqrtf(wave_sum(q) / ((float)n - 1.f));
    if (sd != sd) sd = 0.f;  // single-frame edge case: NaN -> 0 (features.py:90)
    inv = 1.f / (sd + 1e-5f);
  }
  for (int t = lane; t < T; t += 64) st(y + (long long)row * T + t, t < n ? (xr[t] - mu) * inv : pad_value);
}

static int g_logmel_variant = -1;  // -1: not yet read from the environment
static int logmel_variant() {
  if (g_logmel_variant < 0) {
    const char* e = getenv("MI355X_LOGMEL");
    const int v = (e && e[0]) ? atoi(e) : 2;
    g_logmel_variant = v < 0 ? 0 : (v > 2 ? 2 : v);
  }
  return g_logmel_variant;
}
extern "C" int mi355x_logmel_config(int variant) {
  const int old = logmel_variant();
  if (variant >= 0) g_logmel_variant = variant > 2 ? 2 : variant;
  return old;
}
extern "C" int mi355x_logmel_fwd(const void* audio, const void* audio_len, const void* window, int win, int hop, int n_fft,
                                 const void* fb_start, const void* fb_len, const void* fb_off, const void* fb_w, int n_mels,
                                 float preemph, float dither, unsigned seed, float log_guard, void* out, int B, int S, int T,
                                 void* stream) {
  mi_clear_errors();
  if (!audio || !audio_len || !window || !fb_start || !fb_len || !fb_off || !fb_w || !out) return MI_ERR_ARG;
  if (n_fft != NFFT || win <= 0 || win > NFFT || hop <= 0 || hop > NFFT || n_mels <= 0 || B <= 0 || S <= 0) return MI_ERR_ARG;
  if (T != 1 + S / hop) return MI_ERR_ARG;
  const int seg = (MEL_FR - 1) * hop + NFFT;
  // MI355X_LOGMEL / mi355x_logmel_config: 2 (default) = the register-FFT kernel (16 lanes per frame); 1 = the wave-synchronised
  // radix-4 kernel, two frames in flight per wave; 0 = the round-1 kernel (A/B, tests).  An odd hop would misalign the register
  // kernel's 8-byte frame reads: it takes the radix-4 kernel.
  int variant = logmel_variant();
  if (variant == 2 && ((hop & 1) || seg + 1 > 4 * 4 * R16_XP * 2)) variant = 1;
  constexpr int FPW = 2;
  const size_t common = ((seg + 3) & ~3) + NFFT + 2 * NH + (size_t)n_mels * (MEL_FR + 1);
  size_t extra;
  if (variant == 2) extra = 4 * 4 * R16_XP * 2 + FB_CAP + 4 + 3 * (size_t)n_mels;  // (the staging image of seg + 1 floats lives in the transpose buffers)
  else if (variant == 1) extra = 4 * FPW * NH * 2 + 4 * FPW * 260 + FB_CAP + 4 + 3 * (size_t)n_mels;
  else extra = 4 * 2 * NH * 2 + 4 * 260;
  const size_t shm = sizeof(float) * (common + extra);
  if (shm > 160 * 1024) return MI_ERR_ARG;
  dim3 grid((T + MEL_FR - 1) / MEL_FR, B), block(256);
  const void* fn = variant == 2 ? (const void*)logmel_r16_kernel : variant ? (const void*)logmel_wave_kernel<FPW> : (const void*)logmel_kernel;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) {
    (void)hipGetLastError();
    return MI_ERR_LAUNCH;
  }
#define LOGMEL_ARGS (const float*)audio, (const long long*)audio_len, (const float*)window, win, hop, (const int*)fb_start, \
    (const int*)fb_len, (const int*)fb_off, (const float*)fb_w, n_mels, preemph, dither, seed, log_guard, (float*)out, B, S, T
  if (variant == 2) MI_LAUNCH(logmel_r16_kernel, grid, block, shm, (hipStream_t)stream, LOGMEL_ARGS);
  else if (variant) MI_LAUNCH(logmel_wave_kernel<FPW>, grid, block, shm, (hipStream_t)stream, LOGMEL_ARGS);
  else MI_LAUNCH(logmel_kernel, grid, block, shm, (hipStream_t)stream, LOGMEL_ARGS);
#undef LOGMEL_ARGS
  return mi_check_launch();
}

extern "C" int mi355x_feat_normalize(const void* x, const void* seq_len, void* y, int y_dt, int B, int n_mels, int T,
                                     int normalize, float pad_value, void* stream) {
  mi_clear_errors();
  if (!x || !seq_len || !y || B <= 0 || n_mels <= 0 || T <= 0) return MI_ERR_ARG;
  const int rows = B * n_mels;
  hipStream_t s = (hipStream_t)stream;
  if (y_dt == MI_DT_F32)
    MI_LAUNCH((feat_norm_kernel<float>), dim3((rows + 3) / 4), dim3(256), 0, s, (const float*)x,
                       (const long long*)seq_len, (float*)y, rows, n_mels, T, normalize, pad_value);
  else
    MI_LAUNCH((feat_norm_kernel<bf16_t>), dim3((rows + 3) / 4), dim3(256), 0, s, (const float*)x,
                       (const long long*)seq_len, (bf16_t*)y, rows, n_mels, T, normalize, pad_value);
  return mi_check_launch();
}
