"""Flat parameter / gradient storage for the drop-in modules.

Every trainable parameter of a module tree stays an ordinary `nn.Parameter` with the reference's name and shape (the
`.nemo` state-dict ABI), but its storage is a view into ONE contiguous fp32 buffer and its `.grad` a view into ONE
contiguous fp32 gradient buffer.  That is what lets the MI355X path run a single fused AdamW launch per step, zero the
gradients with one fill, write wgrad results straight into their final location, and all-reduce gradients bucket by
bucket over RCCL without a gather copy (reference: 646+ separate tensors through torch DDP buckets / torch AdamW).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
from torch import nn

ALIGN = 64  # elements; keeps every parameter 256-B aligned for the 16-B vector loads of the kernels


class FlatParams:
    def __init__(self, module: nn.Module, tail=None):
        """`tail(name) -> bool` selects parameters that are laid out AFTER all others, contiguously in module order
        (e.g. one equally-shaped weight per layer whose gradients are produced by one batched GEMM at the end of
        backward); `range_of` ignores them, `tail_range` covers them."""
        self.module = module
        self._tail = tail or (lambda name: False)
        self.flat: torch.Tensor = None
        self.grad: torch.Tensor = None
        self.offsets: Dict[str, Tuple[int, int]] = {}
        self.order: List[str] = []
        self.generation = 0  # bumped on every (re)build: dependants (packed weight plans, optimizer state) key on it

    def _params(self):
        ps = [(n, p) for n, p in self.module.named_parameters() if p.dtype == torch.float32]
        return [x for x in ps if not self._tail(x[0])] + [x for x in ps if self._tail(x[0])]

    def is_valid(self) -> bool:
        if self.flat is None:
            return False
        base, gbase = self.flat.data_ptr(), self.grad.data_ptr()
        for n, p in self._params():
            off = self.offsets.get(n)
            if off is None or p.data_ptr() != base + off[0] * 4:
                return False
            if p.grad is None or p.grad.data_ptr() != gbase + off[0] * 4:
                return False
        return True

    def build(self, device=None) -> None:
        params = self._params()
        if not params:
            raise ValueError("module has no fp32 parameters")
        device = device or params[0][1].device
        total = 0
        self.offsets, self.order = {}, []
        for n, p in params:
            self.offsets[n] = (total, p.numel())
            self.order.append(n)
            total += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        flat = torch.zeros(total, dtype=torch.float32, device=device)
        grad = torch.zeros(total, dtype=torch.float32, device=device)
        with torch.no_grad():
            for n, p in params:
                off, num = self.offsets[n]
                view = flat[off: off + num].view(p.shape)
                view.copy_(p.data.to(device))
                gview = grad[off: off + num].view(p.shape)
                if p.grad is not None:
                    gview.copy_(p.grad.to(device))
                p.data = view
                p.grad = gview
        self.flat, self.grad = flat, grad
        self.generation += 1

    def ensure(self, device=None) -> None:
        if not self.is_valid():
            self.build(device)

    def range_of(self, prefix: str) -> Tuple[int, int]:
        """[start, end) element range (aligned) covering every parameter whose name starts with `prefix`."""
        names = [n for n in self.order if n.startswith(prefix) and not self._tail(n)]
        if not names:
            raise KeyError(prefix)
        start = min(self.offsets[n][0] for n in names)
        end = max((self.offsets[n][0] + self.offsets[n][1] + ALIGN - 1) // ALIGN * ALIGN for n in names)
        return start, end

    def tail_range(self) -> Tuple[int, int]:
        names = [n for n in self.order if self._tail(n)]
        if not names:
            return 0, 0
        start = min(self.offsets[n][0] for n in names)
        end = max((self.offsets[n][0] + self.offsets[n][1] + ALIGN - 1) // ALIGN * ALIGN for n in names)
        return start, end

    def trainable_ranges(self) -> List[Tuple[int, int]]:
        """merged, aligned [start, end) element ranges of the parameters with requires_grad=True.  torch.optim.AdamW skips
        parameters without a gradient (frozen ones): the fused optimizer steps these ranges only, so neither the update nor
        the decoupled weight decay touches a frozen parameter.  One range (= one launch) when nothing is frozen."""
        # called per optimizer step (and per reduced bucket): the answer is cached on (generation, the requires_grad flags of a
        # cached parameter list) -- reading 646 flags costs microseconds, walking named_parameters() the module tree every time
        plist = getattr(self, "_plist", None)
        if plist is None or plist[0] != self.generation:
            named = dict(self.module.named_parameters())
            plist = self._plist = (self.generation, [named.get(n) for n in self.order])
        flags = tuple(p is not None and p.requires_grad for p in plist[1])
        cached = getattr(self, "_ranges_cache", None)
        if cached is not None and cached[0] == (self.generation, flags):
            return list(cached[1])
        req = dict(zip(self.order, flags))
        out: List[Tuple[int, int]] = []
        for n in self.order:
            if not req.get(n, False):
                continue
            lo = self.offsets[n][0]
            hi = lo + (self.offsets[n][1] + ALIGN - 1) // ALIGN * ALIGN
            if out and out[-1][1] == lo:
                out[-1] = (out[-1][0], hi)
            else:
                out.append((lo, hi))
        self._ranges_cache = ((self.generation, flags), list(out))
        return out

    def zero_grad(self) -> None:
        from . import ops
        ops.fill_f32(self.grad, 0.0)
