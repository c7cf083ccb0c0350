"""Drop-in for `nemo.collections.asr.losses.CTCLoss` (losses/ctc.py:25-82): same ctor (num_classes = blank id,
zero_infinity, reduction in {none, mean, sum, mean_batch, mean_volume}), typed forward(log_probs [B,T,D], targets [B,U],
input_lengths, target_lengths); arithmetic = the HIP alpha/beta kernel (csrc/ctc.hip), gradient produced in the same
launch and handed to autograd."""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from ..core import LabelsType, LengthsType, LogprobsType, LossType, NeuralModule, NeuralType, typecheck


class _CTCFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_probs, targets, input_lengths, target_lengths, blank, zero_infinity):
        lp = log_probs.to(torch.float32).contiguous()
        grad = torch.empty_like(lp) if log_probs.requires_grad else None
        nll = ops.ctc_loss(lp, targets.contiguous(), input_lengths.contiguous(), target_lengths.contiguous(), blank, grad=grad,
                           grad_scale=1.0, zero_infinity=zero_infinity)
        ctx.grad = grad
        ctx.in_dtype = log_probs.dtype
        return nll

    @staticmethod
    def backward(ctx, dnll):
        g = ctx.grad
        ctx.grad = None
        # d loss / d logp = dnll[b] * grad[b]; the per-utterance factor is a [B] vector (1/B for mean_batch)
        ops.row_scale(g, dnll.to(torch.float32).contiguous(), g.shape[0], g.shape[1] * g.shape[2])
        return g.to(ctx.in_dtype), None, None, None, None, None


class CTCLoss(NeuralModule):
    @property
    def input_types(self):
        return {
            "log_probs": NeuralType(("B", "T", "D"), LogprobsType()),
            "targets": NeuralType(("B", "T"), LabelsType()),
            "input_lengths": NeuralType(tuple("B"), LengthsType()),
            "target_lengths": NeuralType(tuple("B"), LengthsType()),
        }

    @property
    def output_types(self):
        return {"loss": NeuralType(elements_type=LossType())}

    def __init__(self, num_classes, zero_infinity=False, reduction="mean_batch"):
        super().__init__()
        self._blank = num_classes
        if reduction not in ["none", "mean", "sum", "mean_batch", "mean_volume"]:
            raise ValueError("`reduction` must be one of [mean, sum, mean_batch, mean_volume]")
        self.config_reduction = reduction
        self.zero_infinity = zero_infinity
        self.blank = self._blank

    @typecheck()
    def forward(self, log_probs, targets, input_lengths, target_lengths):
        input_lengths = input_lengths.long()
        target_lengths = target_lengths.long()
        targets = targets.long()
        nll = _CTCFn.apply(log_probs, targets, input_lengths, target_lengths, self._blank, self.zero_infinity)
        r = self.config_reduction
        if r == "none":
            return nll
        if r == "sum":
            return nll.sum()
        if r == "mean":  # torch semantics: divide each by its target length (clamped to 1), then batch mean
            return (nll / target_lengths.clamp(min=1).to(nll.dtype)).mean()
        if r == "mean_batch":
            return nll.mean()
        return nll.sum() / target_lengths.sum()  # mean_volume
