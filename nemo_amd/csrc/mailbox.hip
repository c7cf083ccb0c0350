// Statistics mailbox: a latency-path all-reduce (SUM, f64) for the SyncBatchNorm exchanges, one kernel launch per exchange.
//
// What it replaces: the 36 small all_reduce calls per training step of torch.nn.SyncBatchNorm under
// `trainer.sync_batchnorm: true` (examples/asr/conf/conformer/conformer_ctc_bpe.yaml:209; the BatchNorm1d of
// nemo/collections/asr/parts/submodules/conformer_modules.py:339, converted by Lightning).  Through a process group those
// 8-KB calls queue on the ONE RCCL stream behind whatever 64-MiB gradient bucket is in flight (~0.4 ms each) and pay the
// collective's launch protocol on top.  Here every rank owns a mailbox in its own HBM, exported to its peers with
// hipIpcGetMemHandle; an exchange is ONE 256-thread workgroup on the compute stream:
//   1. store my [n] f64 values into slot (seq % SLOTS), box `rank`, of EVERY peer's mailbox (xGMI peer stores, write-through),
//   2. system-scope release, then raise my flag (= seq) in every peer's mailbox,
//   3. wait until all `world` flags of that slot in MY mailbox show seq (system-scope acquire),
//   4. sum the `world` boxes in rank order into the caller's buffer -- the same order on every rank, so all ranks hold
//      bit-identical sums (a ring all-reduce does not promise that; for two ranks it equals any all-reduce bit for bit).
// seq lives in the mailbox (device memory) and is advanced by the kernel itself, so the launch carries no per-call host state.
// Slot reuse: a rank that has finished exchange s has seen every peer's flag s, i.e. every peer has LEFT exchange s-1; the
// fastest rank can therefore be at most one exchange ahead of the slowest, and two slots would do (SLOTS = 4).
// A peer that never arrives (a dead rank) would spin this kernel forever: the wait gives up after `timeout_ms` on the 100-MHz
// wall clock, latches an error word in the mailbox and every later exchange returns at once; mi355x_mailbox_status reads it.
#include <string.h>
#include "common.h"
#include "mi355x_asr.h"

#define MB_SLOTS 4
#define MB_MAX_WORLD 64
#define MB_HDR_BYTES 256  // u64 seq, u64 err, padding

namespace {

struct MbView {              // by-value kernel argument
  char* const* peers;        // device array [world]: base address of every rank's mailbox as mapped HERE (peers[rank] = my own)
  int world, rank, n_max;
  unsigned long long timeout_ticks;
};

__device__ __forceinline__ unsigned long long* mb_flag(char* base, int world, int slot, int r) {
  return (unsigned long long*)(base + MB_HDR_BYTES) + (size_t)slot * world + r;
}
__device__ __forceinline__ unsigned long long* mb_box(char* base, int world, int n_max, int slot, int r) {
  const size_t flags = ((size_t)MB_SLOTS * world * 8 + 255) / 256 * 256;
  return (unsigned long long*)(base + MB_HDR_BYTES + flags) + ((size_t)slot * world + r) * (size_t)n_max;
}

__global__ __launch_bounds__(256) void mailbox_exchange_kernel(double* __restrict__ stats, int n, MbView mb) {
  char* self = mb.peers[mb.rank];
  unsigned long long* hdr = (unsigned long long*)self;
  __shared__ unsigned long long s_seq, s_err;
  __shared__ int s_timed_out;
  if (threadIdx.x == 0) {
    s_seq = __hip_atomic_load(hdr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
    s_err = __hip_atomic_load(hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_timed_out = 0;
  }
  __syncthreads();
  const unsigned long long seq = s_seq;
  const double poison = __longlong_as_double(0x7ff8000000000000ll);  // quiet NaN
  if (s_err) {
    // latched: a peer went missing earlier.  The caller must never mistake its LOCAL sums for the reduced ones (SyncBN would
    // silently turn into per-rank BN with diverging replicas): the result is poisoned, the loss of this step is NaN on this
    // rank, and the host-side latch check (mi355x_mailbox_poll, once per step) raises.
    for (int i = threadIdx.x; i < n; i += 256) stats[i] = poison;
    return;
  }
  const int slot = (int)(seq % MB_SLOTS);
  // 1. my values into every rank's mailbox (my own included: one code path)
  for (int p = 0; p < mb.world; ++p) {
    unsigned long long* dst = mb_box(mb.peers[p], mb.world, mb.n_max, slot, mb.rank);
    for (int i = threadIdx.x; i < n; i += 256)
      __hip_atomic_store(dst + i, (unsigned long long)__double_as_longlong(stats[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
  __syncthreads();
  // 2. flags
  if ((int)threadIdx.x < mb.world)
    __hip_atomic_store(mb_flag(mb.peers[threadIdx.x], mb.world, slot, mb.rank), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  // 3. wait for every rank's flag in MY mailbox
  if ((int)threadIdx.x < mb.world) {
    unsigned long long* f = mb_flag(self, mb.world, slot, threadIdx.x);
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
      if (wall_clock64() - t0 > mb.timeout_ticks) {
        __hip_atomic_store(hdr + 1, 1ull + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // 1 + the rank that is missing
        s_timed_out = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  if (s_timed_out) {
    // a stale slot must not be summed and the sequence number must not advance (the next exchange would then pair this rank's
    // slot with the peer's previous one): poison the result, keep `seq`, stay latched
    for (int i = threadIdx.x; i < n; i += 256) stats[i] = poison;
    return;
  }
  __threadfence_system();
  // 4. rank-ordered sum
  for (int i = threadIdx.x; i < n; i += 256) {
    double s = 0.0;
    for (int r = 0; r < mb.world; ++r) {
      const unsigned long long v =
          __hip_atomic_load(mb_box(self, mb.world, mb.n_max, slot, r) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      s += __longlong_as_double((long long)v);
    }
    stats[i] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(hdr, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

struct mi355x_mailbox {
  int world = 0, rank = 0, n_max = 0, alloc_kind = 0;  // alloc_kind: 1 uncached, 2 fine-grained, 3 plain hipMalloc
  size_t bytes = 0;
  char* base = nullptr;
  char* peers[MB_MAX_WORLD] = {};
  bool opened[MB_MAX_WORLD] = {};
  char** dev_peers = nullptr;
  bool table_dirty = true;
  unsigned long long timeout_ticks = 0;
  unsigned long long* host_hdr = nullptr;  // pinned mirror of the header for mi355x_mailbox_poll
  unsigned long long seen[2] = {0, 0};
  hipEvent_t poll_ev = nullptr;
  bool poll_pending = false;
};

static size_t mb_bytes(int world, int n_max) {
  const size_t flags = ((size_t)MB_SLOTS * world * 8 + 255) / 256 * 256;
  return MB_HDR_BYTES + flags + (size_t)MB_SLOTS * world * (size_t)n_max * 8;
}

#define MB_HIP(call)                                   \
  do {                                                 \
    hipError_t e_ = (call);                            \
    if (e_ != hipSuccess) { (void)hipGetLastError(); return 1000 + (int)e_; } \
  } while (0)

extern "C" int mi355x_mailbox_create(int world, int rank, int n_max, int timeout_ms, int mem_kind, mi355x_mailbox** out,
                                     void* handle_out) {
  if (!out || !handle_out || world < 1 || world > MB_MAX_WORLD || rank < 0 || rank >= world || n_max < 1) return MI_ERR_ARG;
  static_assert(sizeof(hipIpcMemHandle_t) == MI355X_MAILBOX_HANDLE_BYTES, "handle size");
  mi_clear_errors();
  mi355x_mailbox* mb = new mi355x_mailbox();
  mb->world = world; mb->rank = rank; mb->n_max = n_max;
  mb->bytes = mb_bytes(world, n_max);
  mb->timeout_ticks = (unsigned long long)(timeout_ms > 0 ? timeout_ms : 2000) * 100000ull;  // wall_clock64: 100 MHz
  // flags and boxes are written by OTHER devices while a kernel of this one polls them: memory the local L2 does not keep
  // private copies of (uncached, else fine-grained); plain device memory last (same-device peers: the 2-process rehearsal).
  // A kind whose allocation cannot be exported (hipIpcGetMemHandle) is skipped.  mem_kind = 1 | 2 | 3 forces one (0: first that works).
  if (mem_kind < 0 || mem_kind > 3) { delete mb; return MI_ERR_ARG; }
  const int only = mem_kind;
  void* p = nullptr;
  hipIpcMemHandle_t h;
  hipError_t e = hipErrorOutOfMemory;
  for (int kind = 1; kind <= 3 && !mb->alloc_kind; ++kind) {
    if (only && kind != only) continue;
    p = nullptr;
    e = kind == 1 ? hipExtMallocWithFlags(&p, mb->bytes, hipDeviceMallocUncached)
        : kind == 2 ? hipExtMallocWithFlags(&p, mb->bytes, hipDeviceMallocFinegrained)
                    : hipMalloc(&p, mb->bytes);
    if (e == hipSuccess) e = hipMemset(p, 0, mb->bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
    if (e == hipSuccess) { mb->alloc_kind = kind; break; }
    (void)hipGetLastError();
    if (p) (void)hipFree(p);
    p = nullptr;
  }
  if (mb->alloc_kind) e = hipMalloc((void**)&mb->dev_peers, sizeof(char*) * MB_MAX_WORLD);
  if (!mb->alloc_kind || e != hipSuccess) {
    (void)hipGetLastError();
    if (p) (void)hipFree(p);
    delete mb;
    return e == hipSuccess ? MI_ERR_LAUNCH : 1000 + (int)e;
  }
  mb->base = (char*)p;
  memcpy(handle_out, &h, sizeof(h));
  mb->peers[rank] = mb->base;
  mb->opened[rank] = false;
  *out = mb;
  return MI_OK;
}

extern "C" int mi355x_mailbox_open(mi355x_mailbox* mb, int peer, const void* handle) {
  if (!mb || !handle || peer < 0 || peer >= mb->world) return MI_ERR_ARG;
  if (peer == mb->rank) return MI_OK;  // (a process cannot open its own handle; peers[rank] is the allocation itself)
  if (mb->opened[peer]) return MI_ERR_ARG;
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  void* p = nullptr;
  MB_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
  mb->peers[peer] = (char*)p;
  mb->opened[peer] = true;
  mb->table_dirty = true;
  return MI_OK;
}

extern "C" int mi355x_mailbox_exchange(mi355x_mailbox* mb, void* stats_f64, int n, void* stream) {
  if (!mb || !stats_f64 || n < 1 || n > mb->n_max) return MI_ERR_ARG;
  for (int r = 0; r < mb->world; ++r)
    if (!mb->peers[r]) return MI_ERR_ARG;  // a peer's mailbox has not been opened
  hipStream_t s = (hipStream_t)stream;
  mi_clear_errors();
  if (mb->table_dirty) {
    MB_HIP(hipMemcpy(mb->dev_peers, mb->peers, sizeof(char*) * mb->world, hipMemcpyHostToDevice));
    mb->table_dirty = false;
  }
  MbView v{mb->dev_peers, mb->world, mb->rank, mb->n_max, mb->timeout_ticks};
  MI_LAUNCH(mailbox_exchange_kernel, dim3(1), dim3(256), 0, s, (double*)stats_f64, n, v);
  return mi_check_launch();
}

extern "C" int mi355x_mailbox_status(mi355x_mailbox* mb, long long* out3) {
  if (!mb || !out3) return MI_ERR_ARG;
  unsigned long long hdr[2] = {0, 0};
  MB_HIP(hipMemcpy(hdr, mb->base, sizeof(hdr), hipMemcpyDeviceToHost));  // (blocking: diagnostics and tests only)
  out3[0] = (long long)hdr[0];   // exchanges completed
  out3[1] = (long long)hdr[1];   // 0, or 1 + the rank whose flag never arrived
  out3[2] = mb->alloc_kind;
  return MI_OK;
}

// Non-blocking view of the header for the training loop: enqueues a 16-byte copy of (exchanges completed, latch) into pinned host
// memory on `stream` and reports what the PREVIOUS poll's copy brought back (out3[3] = 1 when that copy has landed, else the values
// are those of the poll before).  One call per optimizer step costs one tiny async copy and one event query, never a device sync.
extern "C" int mi355x_mailbox_poll(mi355x_mailbox* mb, void* stream, long long* out4) {
  if (!mb || !out4) return MI_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (!mb->host_hdr) {
    MB_HIP(hipHostMalloc((void**)&mb->host_hdr, 2 * sizeof(unsigned long long), hipHostMallocDefault));
    mb->host_hdr[0] = mb->host_hdr[1] = 0;
    MB_HIP(hipEventCreateWithFlags(&mb->poll_ev, hipEventDisableTiming));
  }
  int fresh = 0;
  if (mb->poll_pending) {
    hipError_t q = hipEventQuery(mb->poll_ev);
    if (q == hipSuccess) {
      mb->seen[0] = mb->host_hdr[0]; mb->seen[1] = mb->host_hdr[1];
      mb->poll_pending = false;
      fresh = 1;
    } else if (q != hipErrorNotReady) {
      (void)hipGetLastError();
      return 1000 + (int)q;
    } else {
      (void)hipGetLastError();
    }
  }
  if (!mb->poll_pending) {
    MB_HIP(hipMemcpyAsync(mb->host_hdr, mb->base, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    MB_HIP(hipEventRecord(mb->poll_ev, s));
    mb->poll_pending = true;
  }
  out4[0] = (long long)mb->seen[0];
  out4[1] = (long long)mb->seen[1];
  out4[2] = mb->alloc_kind;
  out4[3] = fresh;
  return MI_OK;
}

extern "C" void mi355x_mailbox_destroy(mi355x_mailbox* mb) {
  if (!mb) return;
  if (mb->host_hdr) { (void)hipEventSynchronize(mb->poll_ev); (void)hipEventDestroy(mb->poll_ev); (void)hipHostFree(mb->host_hdr); }
  for (int r = 0; r < mb->world; ++r)
    if (mb->opened[r] && mb->peers[r]) (void)hipIpcCloseMemHandle(mb->peers[r]);
  if (mb->dev_peers) (void)hipFree(mb->dev_peers);
  if (mb->base) (void)hipFree(mb->base);
  (void)hipGetLastError();
  delete mb;
}
