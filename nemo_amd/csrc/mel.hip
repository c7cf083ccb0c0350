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

extern "C" int mi355x_logmel_fwd(const void* audio, const void* audio_len, const void* window, int win, int hop, int n_fft,
                                 const void* fb_start, const void* fb_len, const void* fb_off, const void* fb_w, int n_mels,
                                 float preemph, float dither, unsigned seed, float log_guard, void* out, int B, int S, int T,
                                 void* stream) {
  mi_clear_errors();
  if (!audio || !audio_len || !window || !fb_start || !fb_len || !fb_off || !fb_w || !out) return MI_ERR_ARG;
  if (n_fft != NFFT || win <= 0 || win > NFFT || hop <= 0 || hop > NFFT || n_mels <= 0 || B <= 0 || S <= 0) return MI_ERR_ARG;
  if (T != 1 + S / hop) return MI_ERR_ARG;
  const int seg = (MEL_FR - 1) * hop + NFFT;
  const size_t shm = sizeof(float) * (((seg + 3) & ~3) + NFFT + 2 * NH + 4 * 2 * NH * 2 + 4 * 260 + (size_t)n_mels * (MEL_FR + 1));
  if (shm > 160 * 1024) return MI_ERR_ARG;
  dim3 grid((T + MEL_FR - 1) / MEL_FR, B), block(256);
  if (hipFuncSetAttribute((const void*)logmel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) {
    (void)hipGetLastError();
    return MI_ERR_LAUNCH;
  }
  MI_LAUNCH(logmel_kernel, grid, block, shm, (hipStream_t)stream, (const float*)audio, (const long long*)audio_len,
                     (const float*)window, win, hop, (const int*)fb_start, (const int*)fb_len, (const int*)fb_off,
                     (const float*)fb_w, n_mels, preemph, dither, seed, log_guard, (float*)out, B, S, T);
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
