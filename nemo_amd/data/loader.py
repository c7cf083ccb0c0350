"""Host -> HBM hand-over of the padded batches, overlapped with the previous step's compute (SURVEY.md section 8f row 5:
"pinned-memory H2D of padded audio overlapping compute").

The reference leaves this to `torch.utils.data.DataLoader(pin_memory=True)` + Lightning's `batch_to_device`, i.e. a
synchronous copy on the compute stream at the top of each step.  Here a `DeviceBatchLoader` wraps any iterable of collated
host batches: a worker thread pulls batch k+1 (running the dataset / collate code), stages it in reusable page-locked
buffers and enqueues the copies on a dedicated HIP stream while step k computes; `__next__` hands out device tensors after
making the compute stream wait for the copy event -- no host synchronisation anywhere.  A 32 x 20 s fp32 batch is 41 MB:
≈ 0.7 ms of PCIe gen5 time, hidden completely behind a ≈ 47 ms step.
"""
from __future__ import annotations

import queue
import threading
from typing import Iterable, Iterator, List, Optional

import torch


class _Slot:
    """page-locked staging buffers of one in-flight batch, grown on demand and reused"""

    def __init__(self):
        self.bufs: List[Optional[torch.Tensor]] = []
        self.event: Optional[torch.cuda.Event] = None

    def stage(self, i: int, t: torch.Tensor) -> torch.Tensor:
        while len(self.bufs) <= i:
            self.bufs.append(None)
        b = self.bufs[i]
        if b is None or b.dtype != t.dtype or b.numel() < t.numel():
            b = torch.empty(max(t.numel(), 1), dtype=t.dtype).pin_memory()
            self.bufs[i] = b
        v = b[: t.numel()].view(t.shape)
        v.copy_(t)
        return v


class DeviceBatchLoader:
    def __init__(self, batches: Iterable, device, prefetch: int = 2, length_fields=(1,)):
        """length_fields: positions in a batch tuple whose host copy travels with the device tensor (`.host_lengths`): the AUDIO length
        vector of the (audio, audio_len, tokens, token_len) batches of this package.  Only these are tagged -- the encoder takes
        a tagged length vector as permission to size a packed launch sequence from the host copy, which pre-empts recorded
        launch sequences; token ids / token lengths must not trigger that."""
        self.length_fields = tuple(length_fields)
        self.batches = batches
        self.device = torch.device(device)
        self.prefetch = max(1, prefetch)
        self.cuda = self.device.type == "cuda"
        if self.cuda:
            from ..streams import private_stream  # (never a stream somebody else may be capturing on)
            self._copy_stream = private_stream(self.device)
        else:
            self._copy_stream = None
        # prefetch + 1 slots: one being filled, `prefetch` queued, and the consumer's batch keeps its own device tensors
        self._slots = [_Slot() for _ in range(self.prefetch + 1)]

    def __len__(self):
        return len(self.batches)

    def _produce(self, q: "queue.Queue", stop: threading.Event):
        try:
            k = 0
            for batch in self.batches:
                if stop.is_set():
                    return
                if not self.cuda:
                    q.put((batch, None))
                    continue
                slot = self._slots[k % len(self._slots)]
                k += 1
                if slot.event is not None:
                    slot.event.synchronize()  # the copy that last read this slot's pinned buffers has finished
                out = []
                with torch.cuda.stream(self._copy_stream):
                    for i, t in enumerate(batch):
                        if torch.is_tensor(t):
                            dt_ = slot.stage(i, t).to(self.device, non_blocking=True)
                            if i in self.length_fields and t.dim() == 1 and not t.is_floating_point() and t.numel() <= 4096:
                                # length vectors keep a host copy with them: the encoder sizes a PACKED launch sequence (the valid
                                # frames of a ragged batch only, ConformerEncoder._packing_plan) from it without a device read-back
                                dt_.host_lengths = t.detach().clone()
                            out.append(dt_)
                        else:
                            out.append(t)
                    ev = torch.cuda.Event()
                    ev.record(self._copy_stream)
                slot.event = ev
                q.put((out, ev))
            q.put((None, None))
        except BaseException as e:  # surfaced in the consumer
            q.put((e, "error"))

    def __iter__(self) -> Iterator:
        q: "queue.Queue" = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()
        worker = threading.Thread(target=self._produce, args=(q, stop), daemon=True)
        worker.start()
        try:
            while True:
                item, ev = q.get()
                if ev == "error":
                    raise item
                if item is None:
                    return
                if ev is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)
                    for t in item:
                        if torch.is_tensor(t):
                            t.record_stream(cur)  # allocated on the copy stream, consumed on the compute stream
                yield tuple(item)
        finally:
            stop.set()
            while worker.is_alive():  # unblock a producer waiting on the full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    worker.join(timeout=0.01)
