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
"""
from __future__ import annotations

import contextlib
from typing import Callable, List, Optional, Tuple, Union

import torch

_CAPTURE_STREAMS = {}


def capture_stream(device) -> "torch.cuda.Stream":
    """one dedicated capture stream per device (a capture cannot run on the legacy default stream)"""
    key = str(device)
    s = _CAPTURE_STREAMS.get(key)
    if s is None:
        s = _CAPTURE_STREAMS[key] = torch.cuda.Stream(device=device)
    return s


class SegmentedCapture:
    """records [graph | callable]* ; see the module docstring"""

    def __init__(self, device, pool=None):
        self.device = device
        self.pool = pool if pool is not None else torch.cuda.graph_pool_handle()
        self.seq: List[Tuple[str, Union["torch.cuda.CUDAGraph", Callable[[], None]]]] = []
        self._g: Optional["torch.cuda.CUDAGraph"] = None
        self._before_cut: Optional[Callable[[], None]] = None
        self.active = False

    # ---- capture
    def _begin(self):
        self._g = torch.cuda.CUDAGraph()
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
        stream = capture_stream(self.device)
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
        torch.cuda.synchronize(self.device)

    def cut(self, fn: Callable[[], None]) -> None:
        """a step of the sequence that stays a live host call at replay time (collective / hook)"""
        if not self.active:
            raise RuntimeError("cut() outside capturing()")
        self._end()
        self.seq.append(("f", fn))
        self._begin()

    # ---- replay
    def replay(self) -> None:
        for kind, x in self.seq:
            if kind == "g":
                x.replay()
            else:
                x()

    def n_graphs(self) -> int:
        return sum(1 for k, _ in self.seq if k == "g")
