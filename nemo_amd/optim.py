"""Optimizer / LR schedule of the Conformer-CTC recipe on flat buffers.

  * FusedAdamW  -- torch.optim.AdamW semantics (decoupled weight decay, bias correction; the reference builds it through
                   nemo/core/optim/optimizers.py:33 'adamw' with lr 2.0, betas (0.9, 0.98), wd 1e-3,
                   examples/asr/conf/conformer/conformer_ctc_bpe.yaml:175-183) as ONE HIP launch per flat buffer.
  * NoamAnnealing -- nemo/core/optim/lr_scheduler.py:518-576 (same formula, min_lr after warm-up).
"""
from __future__ import annotations

from typing import List

import torch

from . import ops
from .flat import FlatParams


class NoamAnnealing:
    def __init__(self, base_lr: float, d_model: int, warmup_steps=None, warmup_ratio=None, max_steps=None, min_lr=0.0):
        assert not (warmup_steps is not None and warmup_ratio is not None), "Either use particular number of step or ratio"
        assert warmup_ratio is None or max_steps is not None, "If there is a ratio, there should be a total steps"
        if base_lr < min_lr:
            raise ValueError("initial learning rate lower than the minimum learning rate")
        self._normalize = d_model ** (-0.5)
        self.base_lr, self.min_lr = base_lr, min_lr
        self.warmup_steps = warmup_steps if warmup_steps is not None else (int(warmup_ratio * max_steps) if warmup_ratio else 0)
        self.last_epoch = 0

    def lr_at(self, step: int) -> float:
        step = max(1, step)
        if self.warmup_steps > 0:
            mult = self._normalize * min(step ** (-0.5), step * (self.warmup_steps ** (-1.5)))
        else:
            mult = self._normalize * step ** (-0.5)
        out = self.base_lr * mult
        if step > self.warmup_steps:
            out = max(out, self.min_lr)
        return out

    def step(self) -> float:
        self.last_epoch += 1
        return self.lr_at(self.last_epoch)

    def get_last_lr(self) -> float:
        return self.lr_at(self.last_epoch)


class FusedAdamW:
    def __init__(self, flats: List[FlatParams], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.flats = flats
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self._state = {}

    def _moments(self, fp: FlatParams):
        key = (id(fp), fp.generation)
        st = self._state.get(key)
        if st is None:
            st = (torch.zeros_like(fp.flat), torch.zeros_like(fp.flat))
            self._state = {k: v for k, v in self._state.items() if k[0] != id(fp)}
            self._state[key] = st
        return st

    def zero_grad(self):
        for fp in self.flats:
            fp.ensure()
            fp.zero_grad()

    def step(self, lr: float = None, grad_scale: float = 1.0):
        self.step_count += 1
        lr = self.lr if lr is None else lr
        for fp in self.flats:
            m, v = self._moments(fp)
            ops.adamw_step(fp.flat, fp.grad, m, v, lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                           self.step_count, grad_scale)
        for fp in self.flats:
            if hasattr(fp.module, "weights_updated"):
                fp.module.weights_updated()

    def state_dict(self):
        return {"step": self.step_count, "moments": [tuple(t.clone() for t in self._moments(fp)) for fp in self.flats]}
