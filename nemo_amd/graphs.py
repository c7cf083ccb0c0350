"""Replayable launch sequences: a module's forward / backward launch sequence recorded ONCE as hipGraph segments and replayed
every step, so that the host issues a handful of graph launches instead of ~1 000 kernel launches through Python.

The reference's analogue is whole-step CUDA-graph capture (nemo/utils/callbacks/cuda_graph.py:251 `CUDAGraphCallback`); here the
unit is the encoder's hand-sequenced forward / backward (nemo_amd/modules/conformer_encoder.py), because that is where the
launches are, and the sequence is cut into SEGMENTS at every step that must stay a live host call:

  * collectives (SyncBatchNorm statistics, the bucketed gradient all-reduce launched from `grad_ready_hook`) -- RCCL calls
    stay ordinary stream-ordered calls between two graph launches, nothing about the communicator is frozen into a graph;
  * user hooks (optimizer-in-backward).

During capture the Python sequencer runs exactly as in eager mode, on a dedicated capture stream; `cut(fn)` closes the current
graph, records `fn` (it is NOT executed: a capture executes nothing, and an extra collective on one rank would deadlock the
others) and opens the next graph.  `replay()` then launches graph, fn, graph, ... on the current stream.  torch is used for what
it is used everywhere in this package: device memory (the graph-private allocator pool keeps every captured address stable) and
streams.  Per-step varying scalars cannot be kernel arguments of a captured launch; the only ones inside the encoder are the
dropout keys, which read a device-side step word instead (`mi355x_set_step_counter`, include/mi355x_asr.h).

Two ways to run a recorded segment again: `CUDAGraph.replay()` (hipGraphLaunch), or -- `tape=True` -- a LAUNCH TAPE
(`mi355x_tape_*`, csrc/tape.hip): the captured graph's nodes re-issued as live launches from one C loop, on the caller's stream
plus the tape's own lane for the weight-gradient side stream.  hipGraph replay measured 3-4 % slower than live launches on the
device timeline (profiles/r3_host_issue.md); the tape keeps the live-launch timeline and the recorded sequence's host cost.
"""
from __future__ import annotations

import contextlib
import gc
import ctypes
from typing import Callable, List, Optional, Tuple, Union

import torch

_CAPTURE_STREAMS = {}


def capture_stream(device) -> "torch.cuda.Stream":
    """one dedicated capture stream per device (a capture cannot run on the legacy default stream)"""
    key = str(device)
    s = _CAPTURE_STREAMS.get(key)
    if s is None:
        from .streams import private_stream  # (a pooled torch stream may also be somebody's copy / side stream)
        s = _CAPTURE_STREAMS[key] = private_stream(device)
    return s


class Tape:
    """a captured graph as a launch tape (csrc/tape.hip).  Keeps the graph object: the tape launches with the argument copies
    that live inside the graph's nodes."""

    def __init__(self, graph: "torch.cuda.CUDAGraph", max_lanes: int = 4):
        from ._lib import lib, check
        self._lib = lib
        self.graph = graph
        self._h = ctypes.c_void_p()
        rc = lib.mi355x_tape_from_graph(ctypes.c_void_p(graph.raw_cuda_graph()), max_lanes, ctypes.byref(self._h))
        if rc == 2:
            raise NotImplementedError("the captured graph holds a node a launch tape cannot re-issue")
        check(rc, "mi355x_tape_from_graph")
        counts = (ctypes.c_int * 6)()
        check(lib.mi355x_tape_info(self._h, counts), "mi355x_tape_info")
        self.info = dict(zip(("kernels", "memsets", "memcpys", "empty", "lanes", "events"), (int(c) for c in counts)))

    def replay(self, join: bool = True) -> None:
        """join=False: the side lanes (the capture's own side streams) are not joined at the end -- see SegmentedCapture.replay"""
        from ._lib import check
        check(self._lib.mi355x_tape_replay(self._h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 1 if join else 0),
              "mi355x_tape_replay")

    def join(self) -> None:
        from ._lib import check
        check(self._lib.mi355x_tape_join(self._h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "mi355x_tape_join")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.mi355x_tape_destroy(h)
            except Exception:  # noqa: BLE001 -- interpreter shutdown
                pass


class SegmentedCapture:
    """records [graph | callable]* ; see the module docstring"""

    def __init__(self, device, pool=None, tape: bool = False):
        self.device = device
        self.tape = bool(tape)
        self.tape_fallbacks = 0   # segments that stayed hipGraph replays (a node type the tape does not re-issue)
        self.pool = pool if pool is not None else torch.cuda.graph_pool_handle()
        self.seq: List[Tuple[str, Union["torch.cuda.CUDAGraph", Callable[[], None]]]] = []
        self._g: Optional["torch.cuda.CUDAGraph"] = None
        self._before_cut: Optional[Callable[[], None]] = None
        self.active = False

    # ---- capture
    def _begin(self):
        # (tape: the hipGraph itself is what the tape is made from -- keep it, and do not pay for an executable graph)
        self._g = torch.cuda.CUDAGraph(keep_graph=True) if self.tape else torch.cuda.CUDAGraph()
        # thread_local: other host threads (the input pipeline's copy thread) keep making ordinary HIP calls during a capture
        self._g.capture_begin(pool=self.pool, capture_error_mode="thread_local")

    def _end(self):
        if self._before_cut is not None:
            self._before_cut()  # side streams forked into the capture must re-join the capturing stream
        self._g.capture_end()
        self.seq.append(("g", self._g))
        self._g = None

    @contextlib.contextmanager
    def capturing(self, before_cut: Optional[Callable[[], None]] = None):
        """with cap.capturing(): run the launch sequence once; it is recorded, not executed"""
        self._before_cut = before_cut
        torch.cuda.synchronize(self.device)
        # What `torch.cuda.graph.__enter__` does, for the same reason: garbage that holds device resources (an earlier recording's
        # graph, its private pool, a tape's events) must be destroyed BEFORE the capture -- a finaliser that frees a pool while this
        # thread is capturing throws inside a destructor and aborts the process (seen once in the closing run of round 6:
        # "Fatal Python error: Aborted ... Garbage-collecting" under ops.gemm inside _capture_forward) -- and the cyclic collector
        # stays off until the capture has ended.
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        stream = capture_stream(self.device)
        if self.tape:
            from ._lib import lib
            lib.mi355x_tape_log_begin(ctypes.c_void_p(stream.cuda_stream))
        try:
            with torch.cuda.stream(stream):
                self._begin()
                self.active = True
                try:
                    yield self
                except BaseException:
                    self.active = False
                    try:  # leave the stream out of capture mode before the exception travels on
                        self._g.capture_end()
                    except Exception:  # noqa: BLE001
                        pass
                    self._g = None
                    raise
                self.active = False
                self._end()
            if self.tape:
                self._make_tapes()
        finally:
            if self.tape:
                lib.mi355x_tape_log_end()
            if gc_was_on:
                gc.enable()
        torch.cuda.synchronize(self.device)

    def _make_tapes(self) -> None:
        for i, (kind, x) in enumerate(self.seq):
            if kind != "g":
                continue
            try:
                self.seq[i] = ("t", Tape(x))
            except NotImplementedError:
                self.tape_fallbacks += 1   # this segment is replayed as a hipGraph (instantiated at its first replay)

    def cut(self, fn: Callable[[], None]) -> None:
        """a step of the sequence that stays a live host call at replay time (collective / hook)"""
        if not self.active:
            raise RuntimeError("cut() outside capturing()")
        self._end()
        self.seq.append(("f", fn))
        self._begin()

    # ---- replay
    def replay(self, join_between: bool = True) -> None:
        """join_between=False (launch tapes only): a capture has to re-join its side streams before every cut, a tape does not --
        its side lanes ARE those streams, so a live call between two segments that orders itself behind them (the optimizer slice
        or the gradient bucket of a layer) sees what it sees behind the live sequencer, and the main chain does not stall at
        every cut until the layer's weight gradients are done.  The sequence as a whole still ends with a join (after its last
        segment the current stream is behind every lane any of its tapes used)."""
        side = None
        for kind, x in self.seq:
            if kind == "f":
                x()
            elif kind == "t":
                x.replay(join=join_between)
                if not join_between and x.info["lanes"] > 1:
                    side = x
            else:
                x.replay()
        if side is not None:
            side.join()  # (tapes of one capture share their side streams: joining the last multi-lane one joins them all)

    def n_graphs(self) -> int:
        """recorded segments (hipGraph replays and launch tapes)"""
        return sum(1 for k, _ in self.seq if k != "f")

    def tape_info(self):
        """None without tapes; else the node / lane / event counts summed over the segments"""
        tapes = [x for k, x in self.seq if k == "t"]
        if not tapes:
            return None
        out = {k: sum(t.info[k] for t in tapes) for k in ("kernels", "memsets", "memcpys", "empty", "events")}
        out["lanes"] = max(t.info["lanes"] for t in tapes)
        out["segments"], out["graph_fallbacks"] = len(tapes), self.tape_fallbacks
        return out
