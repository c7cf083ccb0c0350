"""Data-parallel gradient exchange for the flat gradient buffers: one process per GPU, torch.distributed (backend
"nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference gets this from Lightning `DDPStrategy` -> torch DDP (25 MB buckets, autograd hooks; trainer.strategy: ddp,
examples/asr/conf/conformer/conformer_ctc_bpe.yaml:201).  Here the backward pass is sequenced by our own host code, so
no hooks are needed: when the backward of a layer finishes, its parameters' gradients are final and CONTIGUOUS in the
flat buffer; `GradSync.ready(start, end)` accumulates such ranges into ~bucket_bytes buckets and launches
`all_reduce` for each full bucket on a side HIP stream (event-ordered after the producing kernels), overlapping the
exchange with the rest of backward.  `wait()` joins the side stream before the optimizer.  The sum is turned into a
mean inside the fused AdamW kernel (`grad_scale = 1/world`), so no extra pass over the gradients is made.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): a few large buckets beat many small ones, hence the 64 MiB
default (121.5 M fp32 gradients = 486 MB -> 8 buckets).

The LAST bucket of a step is the only one nothing can hide: backward is over when it is launched, and the optimizer waits for it.
Ranges arrive in reverse-layer order, so the exchange knows how much of the buffer is still to come (`numel - seen`): when that
remainder drops to `tail_bytes` (16 MiB: the front end's + sub-sampling's gradients, which backward produces last) the pending
ranges are flushed EARLY -- the exposed tail is then one <= 16-MiB all-reduce (+ its AdamW slice) instead of whatever was left of a
64-MiB bucket; and a buffer whose last range has arrived (the decoder's, 66 k values, ready at the very START of backward) is
flushed at once instead of at `wait()`.

`wire_dtype=torch.bfloat16` (MI355X_GRAD_WIRE=bf16) halves the bytes on the links (243 MB per step for Large): a bucket is scaled
by 1/world and rounded to bf16 into a staging buffer on the exchange stream, the bf16 buffer is all-reduced, and the sum -- already
the mean -- is widened back over the fp32 gradient slice (two extra HBM passes over the bucket, both on the exchange stream, i.e.
off the backward chain).  The reference's DDP has the same option as the `bf16_compress_hook` communication hook; like it, this
changes the gradient by one bf16 rounding per rank and is therefore OFF by default.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, grad: torch.Tensor, bucket_bytes: int = 64 << 20, group=None, use_side_stream: Optional[bool] = None,
                 wire_dtype: Optional[torch.dtype] = None, tail_bytes: Optional[int] = None):
        import os
        self.grad = grad
        if tail_bytes is None:
            tail_bytes = int(os.environ.get("MI355X_GRAD_TAIL_BYTES", str(16 << 20)))
        self.tail_elems = max(0, tail_bytes // grad.element_size())
        self._seen = 0          # elements reported ready in this step
        self._tail_cut = False  # the early flush in front of the tail happened (once per step)
        if wire_dtype is None and os.environ.get("MI355X_GRAD_WIRE", "").lower() in ("bf16", "bfloat16"):
            wire_dtype = torch.bfloat16
        if wire_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError("wire_dtype: float32 (default) or bfloat16")
        self.wire_dtype = torch.bfloat16 if wire_dtype == torch.bfloat16 else None
        self._staging = None
        self.bucket_elems = max(1, bucket_bytes // grad.element_size())
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._pending: List[Tuple[int, int]] = []
        self._pending_elems = 0
        self._works = []
        self._reduced: List[Tuple[int, int]] = []
        cuda = grad.is_cuda
        self.use_side_stream = cuda if use_side_stream is None else (use_side_stream and cuda)
        # (a pooled torch stream on purpose: the RCCL process group orders itself against torch's current stream, and that pairing is
        #  the one the collectives have always run with; nemo_amd/streams.py is for the streams that may meet a stream capture)
        self._stream = torch.cuda.Stream(device=grad.device) if self.use_side_stream else None
        # streams other than the current one that also write gradients (the encoder's weight-gradient stream): the
        # exchange stream waits for them too, the backward chain itself never does
        self.producer_streams = lambda: []
        # called as after_reduce(start, end) on the exchange stream right after a range's all-reduce has been enqueued
        # (stream-ordered behind it): the optimizer updates that slice while backward continues
        self.after_reduce = None
        # diagnostics (bench.py): number of all-reduce launches of the last step, and -- with `profile` -- a pair of events
        # around the compute stream's wait for the exchange stream = the exposed (non-overlapped) exchange time
        self.launches_last_step = 0
        self._launches = 0
        self.profile = False
        self.exposed_events = None
        self.bucket_events = []  # with `profile`: (elements, start event, end event) per all-reduce, on the stream that carries it
        self.bucket_events_last_step = []

    # ---- called by the backward sequencer (ranges arrive in reverse-layer order, adjacent ranges are merged)
    def ready(self, start: int, end: int) -> None:
        if self.world <= 1 or end <= start:
            return
        if self._pending and self._pending[-1][0] == end:          # grows downwards (reverse layer order)
            self._pending[-1] = (start, self._pending[-1][1])
        elif self._pending and self._pending[-1][1] == start:
            self._pending[-1] = (self._pending[-1][0], end)
        else:
            self._pending.append((start, end))
        self._pending_elems += end - start
        self._seen += end - start
        left = self.grad.numel() - self._seen
        if left < 0:
            # more elements reported than the buffer holds: ranges arrived twice between two wait() calls (several backward
            # passes per step, or overlapping ranges).  The exchange reduces IN PLACE as ranges arrive, so a second pass would
            # add local gradients onto already-reduced ones, the "everything has arrived" shortcut below would fire on every
            # later range (one tiny all-reduce per layer) and after_reduce would step the optimizer on partial gradients.
            raise RuntimeError("GradSync.ready: ranges covering more than the gradient buffer were reported between two wait() "
                               "calls; accumulate micro-batch gradients before reporting them (one backward per wait()), or call "
                               "wait() after each backward")
        if self._pending_elems >= self.bucket_elems or left <= 0:
            self.flush()
        elif not self._tail_cut and 0 < left <= self.tail_elems and self._pending_elems > self.tail_elems:
            # what is still to come fits the tail bucket: send everything gathered so far now, while backward still runs
            self._tail_cut = True
            self.flush()

    @property
    def grad_scale(self) -> float:
        """what the optimizer still has to multiply the exchanged gradients by: 1/world for the fp32 sum, 1 when the buckets
        travelled as bf16 (they are scaled before the rounding)"""
        return 1.0 if self.wire_dtype is not None else 1.0 / self.world

    def _reduce(self, s: int, e: int):
        """enqueue the all-reduce of grad[s:e] on the current stream; returns the work handle"""
        view = self.grad[s:e]
        if self.wire_dtype is None:
            return dist.all_reduce(view, group=self.group, async_op=True)
        from . import ops
        n = e - s
        if self._staging is None or self._staging.numel() < n:
            self._staging = torch.empty(max(n, self.bucket_elems), dtype=self.wire_dtype, device=self.grad.device)
        # (one staging buffer: the casts and the collective of consecutive buckets are ordered on the exchange stream)
        st = self._staging[:n]
        if n % 8:
            raise RuntimeError("gradient ranges are 64-element aligned (FlatParams.ALIGN)")
        if not view.is_cuda:  # host tensors (the gloo tests of this bookkeeping): the same two casts in torch
            st.copy_(view * (1.0 / self.world))
            work = dist.all_reduce(st, group=self.group, async_op=True)
            work.wait()
            view.copy_(st)
            return work
        ops.drop_scale_cast(view, st, n, 1.0 / self.world)
        work = dist.all_reduce(st, group=self.group, async_op=True)
        work.wait()  # stream-level: the widening copy below is ordered behind the collective, the host does not block
        ops.drop_scale_cast(st, view, n, 1.0)
        return work

    def _timed_reduce(self, s: int, e: int):
        if not (self.profile and self.grad.is_cuda):
            return self._reduce(s, e)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        work = self._reduce(s, e)
        work.wait()  # stream-level (the current stream waits for the collective; the host does not block)
        e1.record()
        self.bucket_events.append((e - s, e0, e1))
        return work

    def flush(self) -> None:
        if self.world <= 1:
            self._pending, self._pending_elems = [], 0
            return
        for (s, e) in self._pending:
            if self._stream is not None:
                self._stream.wait_stream(torch.cuda.current_stream(self.grad.device))
                for ps in self.producer_streams():
                    self._stream.wait_stream(ps)
                with torch.cuda.stream(self._stream):
                    work = self._timed_reduce(s, e)
                    if self.after_reduce is not None:
                        work.wait()  # stream-level: the exchange stream waits for the collective, the host does not
                        self.after_reduce(s, e)
                    else:
                        self._works.append(work)
            else:
                work = self._timed_reduce(s, e)
                if self.after_reduce is not None:
                    work.wait()
                    self.after_reduce(s, e)
                else:
                    self._works.append(work)
            self._reduced.append((s, e))
            self._launches += 1
        self._pending, self._pending_elems = [], 0

    def wait(self) -> float:
        """join all outstanding reductions; returns the factor that turns the sum into the mean"""
        self.flush()
        for w in self._works:
            w.wait()
        self._works = []
        if self._stream is not None:
            cur = torch.cuda.current_stream(self.grad.device)
            if self.profile:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                cur.wait_stream(self._stream)
                e1.record(cur)
                self.exposed_events = (e0, e1)
            else:
                cur.wait_stream(self._stream)
        self._reduced = []
        self.launches_last_step, self._launches = self._launches, 0
        self._seen, self._tail_cut = 0, False
        self.bucket_events_last_step, self.bucket_events = self.bucket_events, []
        return self.grad_scale

    def reduced_ranges(self):
        return list(self._reduced)


class HostSum:
    """Sum of one host-side number over the data-parallel ranks without touching the GPU: a lazily created `gloo` group next
    to the RCCL one (CPU tensors cannot travel over RCCL).  Used for the SyncBatchNorm element count when ranks hold
    different padded lengths: the count is a launch ARGUMENT of the BatchNorm kernels, so it has to be known on the host,
    and a device all-reduce + `.item()` would drain the launch queue every step."""

    def __init__(self):
        self._group = None

    def __call__(self, value: float) -> float:
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return float(value)
        if self._group is None:
            self._group = dist.group.WORLD if dist.get_backend() == "gloo" else dist.new_group(backend="gloo")
        t = torch.tensor([float(value)], dtype=torch.float64)
        dist.all_reduce(t, group=self._group)
        return float(t[0])
