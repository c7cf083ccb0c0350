// Shared device helpers for the MI355X (gfx950 / CDNA4) Conformer-CTC kernels.
// wave = 64 lanes everywhere; no CUDA-compat paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MI_OK 0
#define MI_ERR_ARG 1
#define MI_ERR_LAUNCH 2

#define MI_DT_F32 0
#define MI_DT_BF16 1

typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// Launch-error protocol: entry points call mi_clear_errors() first (hipGetLastError is sticky per host thread, so an
// unrelated earlier failure -- e.g. inside the framework that owns the stream -- must not be blamed on our launch), and
// return 0 or 1000 + hipError_t of their own launch.
static inline void mi_clear_errors() { (void)hipGetLastError(); }
static inline int mi_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MI_OK : 1000 + (int)e;
}

// Every kernel launch of the library goes through MI_LAUNCH.  With mi355x_set_null_launch(1) a launch site issues an EMPTY kernel on
// the same stream instead of its own: the host then pays the same per-launch cost while the GPU has nothing to do, which is how
// tools/host_phases.py measures the pure ISSUE time of a training step (no queue back-pressure from a busy GPU).
// While a launch sequence is being captured for a launch tape (tape.hip: mi355x_tape_log_begin), every launch also notes which
// stream it was captured on -- the one thing a captured graph does not keep, and what the tape's stream lanes are made from.
extern "C" int mi355x_null_launch_flag;
extern int mi355x_tape_log_flag;
void mi_tape_log(hipStream_t stream);
static __global__ void mi_null_kernel() {}
#define MI_LAUNCH(kernel, grid, block, shm, stream, ...)                               \
  do {                                                                                 \
    if (mi355x_null_launch_flag) hipLaunchKernelGGL(mi_null_kernel, dim3(1), dim3(64), 0, stream); \
    else hipLaunchKernelGGL(kernel, grid, block, shm, stream, __VA_ARGS__);            \
    if (mi355x_tape_log_flag) mi_tape_log(stream);                                     \
  } while (0)

// ---------------------------------------------------------------- bf16 <-> f32
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// gfx950 converts in hardware: v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN stays NaN) -- one instruction per pair
// instead of the ~6-op integer sequence per element, which showed up in every bf16-writing epilogue.
typedef __attribute__((ext_vector_type(2))) float mi_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 mi_bf16x2;
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  const mi_f32x2 v = {lo, hi};
  const mi_bf16x2 b = __builtin_convertvector(v, mi_bf16x2);
  return __builtin_bit_cast(uint32_t, b);
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }

// typed load/store helpers (T = float or bf16_t)
template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// 4-wide vector access (16 B for f32, 8 B for bf16); pointers must be aligned accordingly
template <typename T> __device__ __forceinline__ void ld4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void ld4<float>(const float* p, float (&v)[4]) {
  float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void ld4<bf16_t>(const bf16_t* p, float (&v)[4]) {
  uint2 t = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void st4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void st4<float>(float* p, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, const float (&v)[4]) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
}

// ---------------------------------------------------------------- wave / block reductions (wave64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block-wide sum; `red` = __shared__ float[>= blockDim.x/64]; result broadcast to all threads
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// ---------------------------------------------------------------- activations
// v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 VALU instructions): the Swish epilogues are VALU-bound
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }
__device__ __forceinline__ float swish_grad(float x) {  // d/dx x*sigmoid(x)
  float s = sigmoidf_(x);
  return s * (1.f + x * (1.f - s));
}

// ---------------------------------------------------------------- counter-based dropout RNG
// keep(idx) is a pure function of (seed, site, element index): forward epilogues and backward kernels
// regenerate the identical mask, nothing is stored.  2-round 32-bit avalanche hash ("lowbias32" finaliser);
// dropout needs decorrelation, not crypto quality, and Philox4x32-10 per element would cost as much VALU
// time as the FFN GEMM it is fused into.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
struct DropCfg {
  uint32_t key;        // mix of (seed, site)
  uint32_t threshold;  // keep iff rnd >= threshold ; threshold = p * 2^32 ; 0 => dropout off
  float scale;         // 1/(1-p)
  // Device-side step word (mi355x_set_step_counter): when the launch sequence of a training step is replayed from a hipGraph the
  // host cannot pass a fresh seed per step -- the kernel arguments are frozen in the graph -- so every dropout kernel adds
  // the word it finds here to its key at entry (one scalar load).  The host advances the word once per step, before the
  // forward graph; forward and backward of a step therefore regenerate the same masks.  NULL: the key is used as passed.
  const uint32_t* step;
};
// the process-wide step word the launchers put into DropCfg::step (set by mi355x_set_step_counter; defined in version.hip)
extern "C" const uint32_t* mi355x_step_counter_ptr;
static inline DropCfg mi_drop(uint32_t key, uint32_t threshold, float scale) {
  DropCfg d;
  d.key = key; d.threshold = threshold; d.scale = scale;
  d.step = threshold ? mi355x_step_counter_ptr : nullptr;
  return d;
}
// kernel entry: fold the step word into the key (uniform pointer, no store precedes it -> one s_load_dword)
__device__ __forceinline__ void drop_resolve(DropCfg& d) {
  if (d.step) d.key += *d.step;
}
// Elements are hashed in groups of 8 consecutive indices: one strong 32-bit hash per group seeds a xorshift32 stream
// whose (e+1)-th output decides element e.  v_mul_lo_u32 runs at quarter rate on CDNA4, so hashing every element
// (4 multiplies) cost as many VALU cycles as the MFMA main loop of a K=512 GEMM; the xorshift steps are full-rate ops.
__device__ __forceinline__ uint32_t xorshift32(uint32_t s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
__device__ __forceinline__ uint32_t drop_group_seed(const DropCfg& d, uint32_t group) {
  return mix32(mix32(group ^ d.key) + 0x9E3779B9u * (d.key | 1u)) | 1u;  // non-zero state
}
__device__ __forceinline__ float drop_mask(const DropCfg& d, uint32_t idx) {  // scalar form (defines the function)
  if (d.threshold == 0u) return 1.f;
  uint32_t s = drop_group_seed(d, idx >> 3);
  const uint32_t e = idx & 7u;
  for (uint32_t j = 0; j <= e; ++j) s = xorshift32(s);
  return s >= d.threshold ? d.scale : 0.f;
}
// masks of the 8 elements idx8 .. idx8+7 (idx8 % 8 == 0)
__device__ __forceinline__ void drop_mask8(const DropCfg& d, uint32_t idx8, float (&m)[8]) {
  if (d.threshold == 0u) {
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = 1.f;
    return;
  }
  uint32_t s = drop_group_seed(d, idx8 >> 3);
#pragma unroll
  for (int j = 0; j < 8; ++j) { s = xorshift32(s); m[j] = s >= d.threshold ? d.scale : 0.f; }
}
// the same masks for a run that starts in the MIDDLE of a hash group (idx % 8 == 4): a matrix whose width is 4 modulo 8
// (Squeezeformer-Medium: d = 324) starts every other row there.  Elements idx .. idx+3 are outputs 5..8 of their group's stream,
// idx+4 .. idx+7 outputs 1..4 of the next group's -- exactly what drop_mask() defines; any other offset takes the scalar form.
__device__ __forceinline__ void drop_mask8u(const DropCfg& d, uint32_t idx, float (&m)[8]) {
  if (d.threshold == 0u || (idx & 7u) == 0u) { drop_mask8(d, idx, m); return; }
  if ((idx & 7u) == 4u) {
    uint32_t s = drop_group_seed(d, idx >> 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) s = xorshift32(s);
#pragma unroll
    for (int j = 0; j < 4; ++j) { s = xorshift32(s); m[j] = s >= d.threshold ? d.scale : 0.f; }
    s = drop_group_seed(d, (idx >> 3) + 1u);
#pragma unroll
    for (int j = 0; j < 4; ++j) { s = xorshift32(s); m[4 + j] = s >= d.threshold ? d.scale : 0.f; }
    return;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) m[j] = drop_mask(d, idx + j);
}
__device__ __forceinline__ void drop_mask4u(const DropCfg& d, uint32_t idx, float (&m)[4]) {
  if (d.threshold == 0u) {
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = 1.f;
    return;
  }
  const uint32_t o = idx & 7u;
  if (o == 0u || o == 4u) {
    uint32_t s = drop_group_seed(d, idx >> 3);
    if (o) {
#pragma unroll
      for (int j = 0; j < 4; ++j) s = xorshift32(s);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { s = xorshift32(s); m[j] = s >= d.threshold ? d.scale : 0.f; }
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) m[j] = drop_mask(d, idx + j);
}
// standard normal from a counter (Box-Muller on two hashed uniforms) -- dither noise in the mel front-end
__device__ __forceinline__ float hash_normal(uint32_t key, uint32_t idx) {
  uint32_t a = mix32(mix32(idx ^ key) + 0x9E3779B9u);
  uint32_t b = mix32(a ^ 0x85ebca6bU ^ (key * 0xc2b2ae35U));
  float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777217.0f);  // (0,1)
  float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
  return sqrtf(-2.f * __logf(u1)) * __cosf(6.28318530718f * u2);
}

// 16-byte vector I/O of activations, converted to/from f32 registers
template <typename TT> struct VecIO;
template <> struct VecIO<bf16_t> {
  static constexpr int V = 8;
  static __device__ __forceinline__ void load(const bf16_t* p, float* dst) {
    const u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) { dst[2 * j] = __uint_as_float(t[j] << 16); dst[2 * j + 1] = __uint_as_float(t[j] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float* v) {
    u32x4 t = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
    *reinterpret_cast<u32x4*>(p) = t;
  }
};
template <> struct VecIO<float> {
  static constexpr int V = 4;
  static __device__ __forceinline__ void load(const float* p, float* dst) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;
  }
  static __device__ __forceinline__ void store(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};

// Second stage of the two-stage "taps x channels" weight-gradient reductions (depthwise conv, conv1 of the sub-sampling):
//   dw[c*KS + k] += sum_p partial[p][k][c]  (k < KS),   dbias[c] += sum_p partial[p][KS][c]
// grid (ceil((KS+1)*d/256), nsplit): each y-slice sums a contiguous range of parts, so only nsplit-way atomics remain.
static __global__ __launch_bounds__(256) void tap_reduce_kernel(const float* __restrict__ partial, int nparts, int KS, int d,
                                                                float* __restrict__ dw, float* __restrict__ dbias) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= (KS + 1) * d) return;
  const int k = e / d, c = e - k * d;
  const int per = (nparts + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * per, p1 = min(nparts, p0 + per);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const long long ps = (long long)(KS + 1) * d;
  const float* src = partial + (long long)k * d + c;
  int p = p0;
  for (; p + 4 <= p1; p += 4) {
    a0 += src[(p + 0) * ps]; a1 += src[(p + 1) * ps]; a2 += src[(p + 2) * ps]; a3 += src[(p + 3) * ps];
  }
  for (; p < p1; ++p) a0 += src[p * ps];
  const float acc = (a0 + a1) + (a2 + a3);
  if (k < KS) atomicAdd(dw + c * KS + k, acc);
  else if (dbias) atomicAdd(dbias + c, acc);
}

// Second stage of the two-stage column reductions: out[i] += sum_p partial[p*n + i]  (TOut = float or double);
// grid (ceil(n/256), nsplit): nsplit-way atomics only.
template <typename TOut>
static __global__ __launch_bounds__(256) void partials_reduce_kernel(const float* __restrict__ partial, int nparts, int n,
                                                                     TOut* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int per = (nparts + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * per, p1 = min(nparts, p0 + per);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int p = p0;
  for (; p + 4 <= p1; p += 4) {
    a0 += partial[(long long)(p + 0) * n + i]; a1 += partial[(long long)(p + 1) * n + i];
    a2 += partial[(long long)(p + 2) * n + i]; a3 += partial[(long long)(p + 3) * n + i];
  }
  for (; p < p1; ++p) a0 += partial[(long long)p * n + i];
  atomicAdd(out + i, (TOut)((a0 + a1) + (a2 + a3)));
}
