// Launch tapes: a captured launch sequence (hipGraph) re-issued as LIVE launches from one C call.
//
// Why (profiles/r3_host_issue.md, r3_varlen.md): the Python sequencer of the encoder (the analogue of the module calls of
// nemo/collections/asr/modules/conformer_encoder.py:593-759 and of parts/submodules/conformer_modules.py:164-215) needs
// ~19 us of host time per launch -- 16.5 ms per Conformer-CTC-Large step, 22.6 ms per Squeezeformer-Medium step -- and
// variable-length steps are bound by it.  Replaying the recorded hipGraph removes the host cost but runs 3-4 % SLOWER on the
// device timeline on this stack (the graph executor's per-node cost), so neither mode is right for a host-bound step.
//
// A tape takes the DAG out of the captured graph -- kernel nodes with their frozen arguments, memset / memcpy nodes,
// dependencies -- orders it topologically, deals the nodes to stream lanes (lane 0 = the caller's stream; every other stream
// that took part in the capture is a lane again: the weight-gradient side stream comes back as lane 1, ON THAT STREAM, so that
// live calls between two tapes -- optimizer slices, gradient buckets -- that order themselves behind it still do) and
// replays it with hipLaunchKernel / hipMemsetAsync / hipMemcpy3DAsync plus one event per cross-lane edge.  The device sees exactly what the live sequencer
// would have issued; the host pays one C loop (~2-3 us per launch).  Everything a hipGraph replay requires holds here too:
// stable addresses (step arena + graph-private pool), per-step scalars read from device memory (the dropout step word).
//
// The tape does not own the graph: the kernel arguments it launches with live inside the graph's nodes, so the graph must
// outlive the tape (nemo_amd/graphs.py keeps the torch CUDAGraph object next to it).
#include <algorithm>
#include <vector>
#include "common.h"
#include "mi355x_asr.h"

namespace {

enum OpKind { OP_KERNEL = 0, OP_MEMSET = 1, OP_MEMCPY = 2, OP_NOP = 3 };

struct TapeOp {
  int kind = OP_NOP;
  int lane = 0;
  int record = -1;          // event index recorded after the op (it has successors on other lanes), -1: none
  std::vector<int> waits;   // event indices the op's lane waits for before the op
  hipKernelNodeParams k{};
  hipMemsetParams ms{};
  hipMemcpy3DParms mc{};
};

// capture log: graph node -> the stream its launch was captured on (see mi_tape_log)
std::vector<std::pair<hipGraphNode_t, hipStream_t>> g_log;
hipStream_t g_log_origin = nullptr;
bool g_log_sorted = true;

}  // namespace

int mi355x_tape_log_flag = 0;

// called by MI_LAUNCH right after a launch while the log is on: the capturing stream's dependency set is now exactly the node
// this launch created
void mi_tape_log(hipStream_t stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t graph = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t nd = 0;
  if (hipStreamGetCaptureInfo_v2(stream, &st, &id, &graph, &deps, &nd) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  if (st != hipStreamCaptureStatusActive || nd != 1 || !deps) return;
  g_log.emplace_back(deps[0], stream);
  g_log_sorted = false;
}

extern "C" int mi355x_tape_log_begin(void* origin_stream) {
  g_log.clear();
  g_log_sorted = true;
  g_log_origin = (hipStream_t)origin_stream;
  mi355x_tape_log_flag = 1;
  return MI_OK;
}
extern "C" int mi355x_tape_log_end(void) {
  mi355x_tape_log_flag = 0;
  g_log.clear();
  g_log_sorted = true;
  return MI_OK;
}

struct mi355x_tape {
  std::vector<TapeOp> ops;
  std::vector<hipStream_t> side;  // lanes 1..n: the streams of the capture (not owned)
  std::vector<hipEvent_t> ev;     // cross-lane edges
  hipEvent_t ev_fork = nullptr;
  std::vector<hipEvent_t> ev_join;
  int n_kernel = 0, n_memset = 0, n_memcpy = 0, n_nop = 0;
};

#define TAPE_HIP(x)                                   \
  do {                                                \
    hipError_t e_ = (x);                              \
    if (e_ != hipSuccess) {                           \
      (void)hipGetLastError();                        \
      return 1000 + (int)e_;                          \
    }                                                 \
  } while (0)

extern "C" void mi355x_tape_destroy(mi355x_tape* t) {
  if (!t) return;
  for (hipEvent_t e : t->ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : t->ev_join) (void)hipEventDestroy(e);
  if (t->ev_fork) (void)hipEventDestroy(t->ev_fork);
  delete t;
}

// MI_ERR_ARG: bad arguments; 2: the graph holds a node type a tape cannot re-issue (host callbacks, child graphs, external
// events, memory allocations) -- the caller keeps replaying the graph itself; 1000 + hipError_t: a HIP call failed
extern "C" int mi355x_tape_from_graph(void* hip_graph, int max_lanes, mi355x_tape** out) {
  if (!hip_graph || !out || max_lanes < 1) return MI_ERR_ARG;
  *out = nullptr;
  hipGraph_t g = (hipGraph_t)hip_graph;
  if (!g_log_sorted) {
    std::sort(g_log.begin(), g_log.end());
    g_log_sorted = true;
  }
  size_t n = 0;
  TAPE_HIP(hipGraphGetNodes(g, nullptr, &n));
  std::vector<hipGraphNode_t> nodes(n);
  if (n) TAPE_HIP(hipGraphGetNodes(g, nodes.data(), &n));
  nodes.resize(n);
  // node handle -> index (handles are pointers; n is ~1 000: sort + binary search)
  std::vector<std::pair<hipGraphNode_t, int>> index(n);
  for (size_t i = 0; i < n; ++i) index[i] = {nodes[i], (int)i};
  std::sort(index.begin(), index.end());
  auto find = [&](hipGraphNode_t h) -> int {
    auto it = std::lower_bound(index.begin(), index.end(), std::make_pair(h, -1));
    return (it != index.end() && it->first == h) ? it->second : -1;
  };
  std::vector<std::vector<int>> deps(n), succ(n);
  for (size_t i = 0; i < n; ++i) {
    size_t nd = 0;
    TAPE_HIP(hipGraphNodeGetDependencies(nodes[i], nullptr, &nd));
    if (!nd) continue;
    std::vector<hipGraphNode_t> d(nd);
    TAPE_HIP(hipGraphNodeGetDependencies(nodes[i], d.data(), &nd));
    for (size_t j = 0; j < nd; ++j) {
      const int di = find(d[j]);
      if (di < 0) return 2;
      deps[i].push_back(di);
      succ[di].push_back((int)i);
    }
  }
  // topological order, ties by creation index (= the order the sequencer issued the launches in)
  std::vector<int> indeg(n), order;
  order.reserve(n);
  std::vector<int> ready;
  for (size_t i = 0; i < n; ++i) {
    indeg[i] = (int)deps[i].size();
    if (!indeg[i]) ready.push_back((int)i);
  }
  auto cmp = [](int a, int b) { return a > b; };  // min-heap on the creation index
  std::make_heap(ready.begin(), ready.end(), cmp);
  while (!ready.empty()) {
    std::pop_heap(ready.begin(), ready.end(), cmp);
    const int u = ready.back();
    ready.pop_back();
    order.push_back(u);
    for (int v : succ[u])
      if (--indeg[v] == 0) {
        ready.push_back(v);
        std::push_heap(ready.begin(), ready.end(), cmp);
      }
  }
  if (order.size() != n) return 2;  // (a cycle cannot come out of a capture)

  mi355x_tape* t = new mi355x_tape();
  t->ops.resize(n);
  // lanes: the stream every library launch was captured on is in the capture log (mi_tape_log, called by MI_LAUNCH while
  // mi355x_tape_log_begin is in force): the origin stream of the capture is lane 0, every other stream seen gets a lane of
  // its own (the weight-gradient side stream -> lane 1) while lanes are left.  Nodes the log does not know -- the few
  // framework launches and memsets inside the sequence -- go to lane 0; the cross-lane events below keep ANY assignment
  // correct, the lanes only decide what may overlap.
  std::vector<int> lane(n, 0);
  std::vector<hipStream_t> lane_stream{g_log_origin};
  for (size_t i = 0; i < n; ++i) {
    auto it = std::lower_bound(g_log.begin(), g_log.end(), std::make_pair(nodes[i], (hipStream_t) nullptr),
                               [](const std::pair<hipGraphNode_t, hipStream_t>& x, const std::pair<hipGraphNode_t, hipStream_t>& y) {
                                 return x.first < y.first;
                               });
    if (it == g_log.end() || it->first != nodes[i]) continue;
    int l = -1;
    for (size_t k = 0; k < lane_stream.size(); ++k)
      if (lane_stream[k] == it->second) l = (int)k;
    if (l < 0) {
      if ((int)lane_stream.size() >= max_lanes) continue;  // no lane left: lane 0
      l = (int)lane_stream.size();
      lane_stream.push_back(it->second);
    }
    lane[i] = l;
  }
  const int n_lanes = (int)lane_stream.size();
  std::vector<int> pos(n);
  for (size_t k = 0; k < n; ++k) pos[order[k]] = (int)k;
  // events: one per producer with a consumer on another lane.  A lane that has waited for position p of another lane has
  // waited for everything that lane issued before p (stream order), so only later producers need a new wait.
  std::vector<int> ev_of(n, -1);
  std::vector<int> waited((size_t)n_lanes * n_lanes, -1);  // [consumer lane][producer lane] -> latest producer position waited for
  int n_ev = 0;
  for (size_t k = 0; k < n; ++k) {
    const int u = order[k];
    TapeOp& op = t->ops[k];
    op.lane = lane[u];
    for (int pl = 0; pl < n_lanes; ++pl) {
      if (pl == lane[u]) continue;
      int latest = -1;
      for (int d : deps[u])
        if (lane[d] == pl && pos[d] > latest) latest = pos[d];
      int& w = waited[(size_t)lane[u] * n_lanes + pl];
      if (latest <= w) continue;
      w = latest;
      const int d = order[latest];
      if (ev_of[d] < 0) {
        ev_of[d] = n_ev++;
        t->ops[latest].record = ev_of[d];
      }
      op.waits.push_back(ev_of[d]);
    }
    hipGraphNodeType ty;
    hipError_t e = hipGraphNodeGetType(nodes[u], &ty);
    if (e != hipSuccess) { mi355x_tape_destroy(t); return 1000 + (int)e; }
    switch (ty) {
      case hipGraphNodeTypeKernel:
        op.kind = OP_KERNEL;
        e = hipGraphKernelNodeGetParams(nodes[u], &op.k);
        ++t->n_kernel;
        break;
      case hipGraphNodeTypeMemset:
        op.kind = OP_MEMSET;
        e = hipGraphMemsetNodeGetParams(nodes[u], &op.ms);
        ++t->n_memset;
        break;
      case hipGraphNodeTypeMemcpy:
        op.kind = OP_MEMCPY;
        e = hipGraphMemcpyNodeGetParams(nodes[u], &op.mc);
        ++t->n_memcpy;
        break;
      case hipGraphNodeTypeEmpty:
        op.kind = OP_NOP;
        ++t->n_nop;
        break;
      default:
        mi355x_tape_destroy(t);
        return 2;
    }
    if (e != hipSuccess) { (void)hipGetLastError(); mi355x_tape_destroy(t); return e == hipErrorInvalidValue ? 2 : 1000 + (int)e; }
    if (op.kind == OP_KERNEL && (op.k.extra != nullptr || op.k.func == nullptr)) { mi355x_tape_destroy(t); return 2; }
  }
  for (int l = 1; l < n_lanes; ++l) {
    t->side.push_back(lane_stream[l]);
    hipEvent_t ej;
    hipError_t e = hipEventCreateWithFlags(&ej, hipEventDisableTiming);
    if (e != hipSuccess) { mi355x_tape_destroy(t); return 1000 + (int)e; }
    t->ev_join.push_back(ej);
  }
  if (n_lanes > 1) {
    hipError_t e = hipEventCreateWithFlags(&t->ev_fork, hipEventDisableTiming);
    if (e != hipSuccess) { mi355x_tape_destroy(t); return 1000 + (int)e; }
  }
  for (int i = 0; i < n_ev; ++i) {
    hipEvent_t ev;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) { mi355x_tape_destroy(t); return 1000 + (int)e; }
    t->ev.push_back(ev);
  }
  *out = t;
  return MI_OK;
}

extern "C" int mi355x_tape_replay(mi355x_tape* t, void* stream, int join) {
  if (!t) return MI_ERR_ARG;
  hipStream_t main = (hipStream_t)stream;
  const size_t n_side = t->side.size();
  for (hipStream_t s : t->side)
    if (s == main) return MI_ERR_ARG;  // the caller's stream is lane 0; it cannot be a side lane of the same tape as well
  if (n_side) {  // the side lanes start behind whatever the caller's stream already holds
    TAPE_HIP(hipEventRecord(t->ev_fork, main));
    for (hipStream_t s : t->side) TAPE_HIP(hipStreamWaitEvent(s, t->ev_fork, 0));
  }
  for (TapeOp& op : t->ops) {
    hipStream_t s = op.lane == 0 ? main : t->side[op.lane - 1];
    for (int w : op.waits) TAPE_HIP(hipStreamWaitEvent(s, t->ev[w], 0));
    switch (op.kind) {
      case OP_KERNEL:
        TAPE_HIP(hipLaunchKernel(op.k.func, op.k.gridDim, op.k.blockDim, op.k.kernelParams, op.k.sharedMemBytes, s));
        break;
      case OP_MEMSET: {
        const hipMemsetParams& m = op.ms;
        if (m.height <= 1) {
          if (m.elementSize == 1) TAPE_HIP(hipMemsetAsync(m.dst, (int)m.value, m.width, s));
          else if (m.elementSize == 2) TAPE_HIP(hipMemsetD16Async((hipDeviceptr_t)m.dst, (unsigned short)m.value, m.width, s));
          else TAPE_HIP(hipMemsetD32Async((hipDeviceptr_t)m.dst, (int)m.value, m.width, s));
        } else {
          if (m.elementSize != 1) return 2;
          TAPE_HIP(hipMemset2DAsync(m.dst, m.pitch, (int)m.value, m.width, m.height, s));
        }
      } break;
      case OP_MEMCPY:
        TAPE_HIP(hipMemcpy3DAsync(&op.mc, s));
        break;
      default:
        break;
    }
    if (op.record >= 0) TAPE_HIP(hipEventRecord(t->ev[op.record], s));
  }
  // join: the caller's stream continues behind every lane (a graph launch's semantics).  Without it the side lanes run on, as
  // they do behind the live sequencer: whoever consumes their results orders itself behind those streams.
  for (size_t l = 0; join && l < n_side; ++l) {
    TAPE_HIP(hipEventRecord(t->ev_join[l], t->side[l]));
    TAPE_HIP(hipStreamWaitEvent(main, t->ev_join[l], 0));
  }
  return MI_OK;
}

// `stream` continues behind whatever the tape's side lanes hold now (the join a replay with join = 0 left out)
extern "C" int mi355x_tape_join(mi355x_tape* t, void* stream) {
  if (!t) return MI_ERR_ARG;
  hipStream_t main = (hipStream_t)stream;
  for (size_t l = 0; l < t->side.size(); ++l) {
    TAPE_HIP(hipEventRecord(t->ev_join[l], t->side[l]));
    TAPE_HIP(hipStreamWaitEvent(main, t->ev_join[l], 0));
  }
  return MI_OK;
}

// counts[0..5] = kernels, memsets, memcpys, empty nodes, lanes, cross-lane events
extern "C" int mi355x_tape_info(const mi355x_tape* t, int* counts) {
  if (!t || !counts) return MI_ERR_ARG;
  counts[0] = t->n_kernel; counts[1] = t->n_memset; counts[2] = t->n_memcpy; counts[3] = t->n_nop;
  counts[4] = 1 + (int)t->side.size(); counts[5] = (int)t->ev.size();
  return MI_OK;
}

// ---- private streams.  torch hands its streams out of a fixed pool (32 per priority and device), round-robin: the 33rd
// torch.cuda.Stream() of a process IS the first one again.  Two owners that believe they hold different streams -- the capture
// stream of the recorded launch sequences, an encoder's weight-gradient stream, the input pipeline's copy stream -- then share
// one, and a host-to-device copy issued by the loader thread lands inside somebody's stream capture ("capturing stream has
// unjoined work", seen once the GPU suite had grown past 32 stream creations).  Streams created here belong to their owner alone;
// the Python side wraps them as torch.cuda.ExternalStream (nemo_amd/streams.py).
extern "C" int mi355x_stream_create(int priority, void** out) {
  if (!out) return MI_ERR_ARG;
  hipStream_t s = nullptr;
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // lo = least urgent (numerically greatest), hi = most urgent
  int pr = priority < hi ? hi : (priority > lo ? lo : priority);
  TAPE_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, pr));
  *out = (void*)s;
  return MI_OK;
}
extern "C" int mi355x_stream_destroy(void* stream) {
  if (!stream) return MI_ERR_ARG;
  TAPE_HIP(hipStreamDestroy((hipStream_t)stream));
  return MI_OK;
}
