"""Streams that belong to their owner alone.

`torch.cuda.Stream()` hands streams out of a fixed pool (32 per priority and device), round-robin: the 33rd creation in a process
returns the first stream again.  Owners that must NOT share a stream -- the capture stream of the recorded launch sequences
(nemo_amd/graphs.py), an encoder's weight-gradient stream, the input pipeline's copy stream (nemo_amd/data/loader.py), the
gradient exchange stream (nemo_amd/parallel.py) -- therefore create theirs through the library (`mi355x_stream_create`) and use
it as a `torch.cuda.ExternalStream`: torch keeps owning device memory and stream semantics, the handle is nobody else's.
(The reference relies on torch's pool too -- e.g. nemo/utils/callbacks/cuda_graph.py:251 captures on `torch.cuda.Stream()` -- with
one stream user per process; here one process holds several.)
"""
from __future__ import annotations

import ctypes

import torch


_FREE = {}   # (device index, priority) -> handles whose owner is gone


class _Owned(torch.cuda.ExternalStream):
    """an ExternalStream on a library-created HIP stream.  The stream is never destroyed -- torch's caching allocator may still hold
    blocks that were recorded on it (`Tensor.record_stream`) and would query a dead handle -- it goes back to a free list when its
    owner is collected, and the next owner gets it for itself."""

    def __del__(self):
        key, h = getattr(self, "_mi355x_key", None), getattr(self, "_mi355x_handle", None)
        if key is not None and h:
            _FREE.setdefault(key, []).append(h)


def private_stream(device, priority: int = 0) -> "torch.cuda.Stream":
    """a non-blocking HIP stream on `device` that nobody else holds (HIP priority: 0 default, -1 more urgent), wrapped for torch"""
    from ._lib import check, lib
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, int(priority))
    free = _FREE.get(key)
    if free:
        handle = free.pop()
    else:
        out = ctypes.c_void_p()
        with torch.cuda.device(idx):
            check(lib.mi355x_stream_create(int(priority), ctypes.byref(out)), "mi355x_stream_create")
        handle = out.value
    s = _Owned(handle, device=torch.device("cuda", idx))
    s._mi355x_handle, s._mi355x_key = handle, key
    return s
