"""Step-scoped bump allocator for the encoder's hand-sequenced forward / backward.

The sequencer asks for the same ~800 tensors in the same order every step; only their sizes move with the padded length of the
batch.  Through torch's caching allocator a never-seen size is a slow path (measured with variable-length batches: 47 us per
`torch.empty`, 38 ms of host time per step -- `tools/host_profile.py`), and every step of a duration-bucketed run brings
never-seen sizes.  With 288 GB of HBM the simplest allocator wins: one buffer per phase (forward / backward), handed out front
to back, rewound at the start of the next step's phase.  Nothing is freed inside a step, so tensors touched by the weight-gradient
side stream need no stream bookkeeping, and a recorded launch sequence (hipGraph) sees stable addresses for free.

The buffer sizes itself: a cycle that overflows falls back to `torch.empty` for the overflow and the NEXT rewind grows the buffer
to 1.25 x what the cycle asked for (the outgrown buffer is kept alive: a recorded graph may hold addresses inside it).
`torch` stays what it is everywhere in this package -- the owner of device memory.
"""
from __future__ import annotations

from typing import List, Optional

import torch


class Arena:
    def __init__(self, name: str = ""):
        self.name = name
        self.buf: Optional[torch.Tensor] = None
        self.off = 0          # bytes handed out in this cycle
        self.need = 0         # bytes asked for in this cycle (overflow included)
        self.gen = 0          # cycles so far: whoever keeps tensors across a rewind can detect that they are gone
        self.retired: List[torch.Tensor] = []
        self.disabled = False

    def rewind(self, device) -> None:
        cap = self.buf.numel() if (self.buf is not None and self.buf.device == device) else 0
        if self.need > cap and not self.disabled:
            want = int(self.need * 1.25) + (1 << 20)
            try:
                new = torch.empty(want, dtype=torch.uint8, device=device)
            except RuntimeError:  # out of memory: this arena steps aside, the caching allocator takes over for good
                self.disabled, new = True, None
            if new is not None:
                if self.buf is not None:
                    self.retired.append(self.buf)
                self.buf = new
        self.off, self.need = 0, 0
        self.gen += 1

    def take(self, shape, dtype, device) -> torch.Tensor:
        n = 1
        for k in shape:
            n *= int(k)
        nbytes = n * _ITEM[dtype]
        step = (nbytes + 255) & ~255
        self.need += step
        buf = self.buf
        if buf is None or buf.device != device or self.off + step > buf.numel():
            return torch.empty(*shape, dtype=dtype, device=device)
        t = buf[self.off: self.off + nbytes].view(dtype).view(*shape)
        self.off += step
        return t


_ITEM = {torch.float32: 4, torch.bfloat16: 2, torch.float16: 2, torch.float64: 8, torch.int64: 8, torch.int32: 4, torch.uint8: 1,
         torch.int8: 1, torch.bool: 1}
