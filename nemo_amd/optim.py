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
import math

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


class NoamHoldAnnealing:
    """nemo/core/optim/lr_scheduler.py:153-228 (WarmupHoldPolicy.get_lr) + :429-435, 578-639: linear warm-up to the peak lr,
    hold, then lr * warmup^rate / (step - hold)^rate (the Squeezeformer recipe: decay_rate 1.0).  Same step / get_last_lr
    surface as NoamAnnealing."""

    def __init__(self, base_lr: float, warmup_steps=None, warmup_ratio=None, hold_steps=None, hold_ratio=None, max_steps=None,
                 decay_rate=0.5, min_lr=0.0):
        assert not (warmup_steps is not None and warmup_ratio is not None), "Either use particular number of step or ratio"
        assert not (hold_steps is not None and hold_ratio is not None), "Either use particular number of step or ratio"
        assert hold_ratio is None or max_steps is not None, "If there is a ratio, there should be a total steps"
        assert warmup_ratio is None or max_steps is not None, "If there is a ratio, there should be a total steps"
        self.base_lr, self.min_lr, self.decay_rate, self.max_steps = base_lr, min_lr, decay_rate, max_steps
        self.warmup_steps = warmup_steps if warmup_steps is not None else (int(warmup_ratio * max_steps) if warmup_ratio else 0)
        if hold_steps is not None:
            self.hold_steps = hold_steps + self.warmup_steps
        elif hold_ratio is not None:
            self.hold_steps = int(hold_ratio * max_steps) + self.warmup_steps
        else:
            self.hold_steps = 0
        self.last_epoch = 0

    def lr_at(self, step: int) -> float:
        if step <= self.warmup_steps and self.warmup_steps > 0:
            return self.base_lr * (step + 1) / (self.warmup_steps + 1)  # _get_warmup_lr (:138-140)
        if self.warmup_steps <= step < self.hold_steps:
            return self.base_lr
        if self.max_steps is not None and step > self.max_steps:
            return self.min_lr
        if not self.warmup_steps:
            raise ValueError("Noam scheduler cannot be used without warmup steps")
        hold = self.hold_steps - self.warmup_steps if self.hold_steps > 0 else 0
        t_warm = max(1, self.warmup_steps ** self.decay_rate)
        t_hold = max(1, (step - hold) ** self.decay_rate)
        return max(self.base_lr * t_warm / t_hold, self.min_lr)

    def step(self) -> float:
        self.last_epoch += 1
        return self.lr_at(self.last_epoch)

    def get_last_lr(self) -> float:
        return self.lr_at(self.last_epoch)


class CosineAnnealing:
    """nemo/core/optim/lr_scheduler.py:273-385 (WarmupAnnealHoldPolicy.get_lr) + :387-414, 467-515 -- the FastConformer recipes'
    schedule: linear warm-up (step + 1) / (warmup + 1), cosine decay to min_lr at max_steps; with `constant_steps` the
    Megatron variant (warm-up step / warmup, decay over max_steps - warmup - constant, then min_lr)."""

    def __init__(self, base_lr: float, max_steps: int, warmup_steps=None, warmup_ratio=None, constant_steps=None,
                 constant_ratio=None, min_lr=0.0):
        assert not (warmup_steps is not None and warmup_ratio is not None), "Either use particular number of step or ratio"
        assert not (constant_steps is not None and constant_ratio is not None), "Either use constant_steps or constant_ratio"
        if base_lr < min_lr:
            raise ValueError("received an initial learning rate that was lower than the minimum learning rate.")
        self.base_lr, self.min_lr, self.max_steps = base_lr, min_lr, max_steps
        self.warmup_steps = warmup_steps if warmup_steps is not None else (int(warmup_ratio * max_steps) if warmup_ratio else 0)
        self.constant_steps = constant_steps if constant_steps is not None else (
            int(constant_ratio * max_steps) if constant_ratio else 0)
        self.decay_steps = max_steps - (self.constant_steps + self.warmup_steps)
        self.last_epoch = 0

    def _megatron(self, step):
        if self.warmup_steps > 0 and step <= self.warmup_steps:
            return self.base_lr * float(step) / float(self.warmup_steps)
        if step > self.warmup_steps + self.decay_steps:
            return self.min_lr
        ratio = float(step - self.warmup_steps) / float(self.decay_steps)
        return self.min_lr + 0.5 * (math.cos(math.pi * ratio) + 1.0) * (self.base_lr - self.min_lr)

    def lr_at(self, step: int) -> float:
        if self.constant_steps:
            return self._megatron(step) if step <= self.max_steps else self.min_lr
        if self.warmup_steps > 0 and step <= self.warmup_steps:
            return self.base_lr * (step + 1) / (self.warmup_steps + 1)
        if step > self.max_steps:
            return self.min_lr
        mult = 0.5 * (1 + math.cos(math.pi * (step - self.warmup_steps) / (self.max_steps - self.warmup_steps)))
        return (self.base_lr - self.min_lr) * mult + self.min_lr

    def step(self) -> float:
        self.last_epoch += 1
        return self.lr_at(self.last_epoch)

    def get_last_lr(self) -> float:
        return self.lr_at(self.last_epoch)


class FusedAdamW:
    """`max_grad_norm` = trainer.gradient_clip_val (global L2 norm over every flat buffer, computed and applied on the
    device); `ema_decay` = the EMA callback's decay (nemo/collections/common/callbacks/ema.py): the average is updated inside
    the AdamW launch, `swap_ema_weights()` is the reference's context manager of the same name (ema.py:295-318)."""

    def __init__(self, flats: List[FlatParams], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None,
                 ema_decay=None):
        self.flats = flats
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = float(max_grad_norm) if max_grad_norm else None
        if ema_decay is not None and not (0.0 <= ema_decay <= 1.0):
            raise ValueError("EMA decay value must be between 0 and 1")  # ema.py:49-50
        self.ema_decay = ema_decay
        self.step_count = 0
        self._state = {}
        self._ema = {}
        self._clip = None
        self.last_grad_norm = None  # device tensor [1] after a clipped step

    def _moments(self, fp: FlatParams):
        key = (id(fp), fp.generation)
        st = self._state.get(key)
        if st is None:
            old = [v for k, v in self._state.items() if k[0] == id(fp)]
            if old and old[0][0].numel() == fp.flat.numel():
                # the flat buffer was rebuilt (model.to(), zero_grad(set_to_none=True) ...): same parameters, same layout
                # -> the Adam moments move with them instead of silently restarting from zero
                st = tuple(t.to(fp.flat.device) for t in old[0])
            elif old:
                raise RuntimeError("FusedAdamW: the parameter set changed under the optimizer (flat buffer rebuilt with a "
                                   "different size); call setup_optimization() again")
            else:
                st = (torch.zeros_like(fp.flat), torch.zeros_like(fp.flat))
            self._state = {k: v for k, v in self._state.items() if k[0] != id(fp)}
            self._state[key] = st
        return st

    def _ema_of(self, fp: FlatParams):
        if self.ema_decay is None:
            return None
        key = (id(fp), fp.generation)
        e = self._ema.get(key)
        if e is None:
            old = [v for k, v in self._ema.items() if k[0] == id(fp)]
            if old and old[0].numel() == fp.flat.numel():
                e = old[0].to(fp.flat.device)  # buffer rebuilt: the running average is kept
            else:
                e = fp.flat.detach().clone()  # the average starts at the current weights (ema.py:263-266)
            self._ema = {k: v for k, v in self._ema.items() if k[0] != id(fp)}
            self._ema[key] = e
        return e

    def zero_grad(self):
        for fp in self.flats:
            fp.ensure()
            fp.zero_grad()

    # ---- optimizer step in pieces, while backward is still running.  `begin_step` fixes this step's constants;
    # `step_range(fp, lo, hi)` updates one contiguous slice of a flat buffer as soon as its gradients are final (the
    # backward sequencer / the gradient exchange call it on THEIR stream: the weight-gradient stream on one GPU, the RCCL
    # stream after a bucket's all-reduce in data-parallel runs); `finish_step` updates whatever is left.  AdamW is
    # HBM-bound and the rest of backward is MFMA-bound, so the update disappears behind it.  Not used with gradient
    # clipping (the global norm needs every gradient first).
    def begin_step(self, lr: float = None, grad_scale: float = 1.0):
        self.step_count += 1
        self._cur = (self.lr if lr is None else lr, grad_scale)
        self._done = {id(fp): [] for fp in self.flats}
        for fp in self.flats:  # state is created here, on the caller's stream, not on whichever stream steps first
            fp.ensure()
            self._moments(fp)
            self._ema_of(fp)
        return self.max_grad_norm is None

    def step_range(self, fp: FlatParams, lo: int, hi: int):
        if getattr(self, "_cur", None) is None or self.max_grad_norm is not None or hi <= lo:
            return
        lr, grad_scale = self._cur
        m, v = self._moments(fp)
        ema = self._ema_of(fp)
        for tlo, thi in fp.trainable_ranges():  # frozen parameters (requires_grad=False) are never stepped nor decayed
            a, b = max(lo, tlo), min(hi, thi)
            if a >= b:
                continue
            ops.adamw_step(fp.flat[a:b], fp.grad[a:b], m[a:b], v[a:b], lr, self.betas[0], self.betas[1], self.eps,
                           self.weight_decay, self.step_count, grad_scale, ema=None if ema is None else ema[a:b],
                           ema_decay=self.ema_decay or 0.0)
        self._done[id(fp)].append((lo, hi))

    def finish_step(self):
        lr, grad_scale = self._cur
        for fp in self.flats:
            done = sorted(self._done[id(fp)])
            pos, rest = 0, []
            for lo, hi in done:
                if lo > pos:
                    rest.append((pos, lo))
                pos = max(pos, hi)
            if pos < fp.flat.numel():
                rest.append((pos, fp.flat.numel()))
            for lo, hi in rest:
                self.step_range(fp, lo, hi)
        self._cur = None
        for fp in self.flats:
            if hasattr(fp.module, "weights_updated"):
                fp.module.weights_updated()

    def step(self, lr: float = None, grad_scale: float = 1.0):
        self.step_count += 1
        lr = self.lr if lr is None else lr
        coef = None
        if self.max_grad_norm:
            dev = self.flats[0].flat.device
            if self._clip is None or self._clip[0].device != dev:
                self._clip = (torch.zeros(len(self.flats), dtype=torch.float64, device=dev),
                              torch.zeros(2, dtype=torch.float32, device=dev))
            sumsq, coef = self._clip
            sumsq.zero_()
            for i, fp in enumerate(self.flats):  # (clip_grad_norm_ sees the gradients of trainable parameters only)
                for lo, hi in fp.trainable_ranges():
                    ops.grad_sumsq(fp.grad[lo:hi], sumsq[i:i + 1])
            ops.clip_coef(sumsq, grad_scale, self.max_grad_norm, coef)
            self.last_grad_norm = coef[1:2]
        for fp in self.flats:
            m, v = self._moments(fp)
            ema = self._ema_of(fp)
            # torch.optim.AdamW skips parameters whose grad is None: frozen parameters (requires_grad=False, .freeze())
            # receive neither the update nor the decoupled weight decay.  One range = one launch when nothing is frozen.
            for lo, hi in fp.trainable_ranges():
                ops.adamw_step(fp.flat[lo:hi], fp.grad[lo:hi], m[lo:hi], v[lo:hi], lr, self.betas[0], self.betas[1], self.eps,
                               self.weight_decay, self.step_count, grad_scale, clip_coef=coef,
                               ema=None if ema is None else ema[lo:hi], ema_decay=self.ema_decay or 0.0)
        for fp in self.flats:
            if hasattr(fp.module, "weights_updated"):
                fp.module.weights_updated()

    def swap_ema_weights(self):
        """context manager: the modules' parameters hold the EMA weights inside the block (evaluation / checkpointing)"""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            if self.ema_decay is None:
                yield
                return
            pairs = [(fp, self._ema_of(fp)) for fp in self.flats]

            def swap():
                for fp, e in pairs:
                    tmp = fp.flat.detach().clone()
                    fp.flat.copy_(e)
                    e.copy_(tmp)
                    if hasattr(fp.module, "weights_updated"):
                        fp.module.weights_updated()
            swap()
            try:
                yield
            finally:
                swap()
        return ctx()

    def state_dict(self):
        sd = {"step": self.step_count, "moments": [tuple(t.clone() for t in self._moments(fp)) for fp in self.flats]}
        if self.ema_decay is not None:
            sd["ema"] = [self._ema_of(fp).clone() for fp in self.flats]
            sd["decay"] = self.ema_decay
        return sd

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        for fp, (m, v) in zip(self.flats, sd["moments"]):
            fp.ensure()
            mm, vv = self._moments(fp)
            mm.copy_(m); vv.copy_(v)
        if self.ema_decay is not None and "ema" in sd:
            for fp, e in zip(self.flats, sd["ema"]):
                self._ema_of(fp).copy_(e)

