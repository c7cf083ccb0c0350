"""Drop-in for `RNNTLossNumba` (`nemo/collections/asr/parts/numba/rnnt_loss/rnnt_pytorch.py:393-440`, autograd function
`_RNNTNumba` :39-98; selected as loss_name 'warprnnt_numba' by `losses/rnnt.py:88-158`): the joint network's logits
`acts [B,T,U+1,V+1]` go in, the log-softmax is fused into the loss kernels, the gradient w.r.t. the logits is produced in
the forward call and handed out (scaled by the upstream gradient) in backward -- same contract, HIP kernels underneath
(`csrc/rnnt.hip`, C-ABI `mi355x_rnnt_loss`).  There is no CPU path: CPU tensors raise."""
from __future__ import annotations

import torch
from torch import nn

from .. import ops


def certify_inputs(log_probs, labels, lengths, label_lengths):
    """rnnt_numpy.py:55-98 (same messages); the two max() reads are host syncs, as in the reference"""
    for var, t, name in ((labels, torch.int64, "labels"), (label_lengths, torch.int64, "label_lengths"),
                         (lengths, torch.int64, "lengths")):
        if var.dtype is not t:
            raise TypeError("{} must be {}".format(name, t))
    for var, name in ((log_probs, "log_probs"), (labels, "labels"), (label_lengths, "label_lengths"), (lengths, "lengths")):
        if not var.is_contiguous():
            raise ValueError("{} must be contiguous".format(name))
    if lengths.shape[0] != log_probs.shape[0]:
        raise ValueError(f"Must have a length per example. Given lengths dim: {lengths.shape[0]}, "
                         f"Log probs dim : {log_probs.shape[0]}")
    if label_lengths.shape[0] != log_probs.shape[0]:
        raise ValueError("Must have a label length per example. "
                         f"Given label lengths dim : {label_lengths.shape[0]}, Log probs dim : {log_probs.shape[0]}")
    for var, dim, name in ((log_probs, 4, "log_probs"), (labels, 2, "labels"), (lengths, 1, "lenghts"),
                           (label_lengths, 1, "label_lenghts")):
        if len(var.shape) != dim:
            raise ValueError("{} must be {}D".format(name, dim))
    max_T, max_U = torch.max(lengths), torch.max(label_lengths)
    T, U = log_probs.shape[1:3]
    if T != max_T:
        raise ValueError(f"Input length mismatch! Given T: {T}, Expected max T from input lengths: {max_T}")
    if U != max_U + 1:
        raise ValueError(f"Output length mismatch! Given U: {U}, Expected max U from target lengths: {max_U} + 1")


class _RNNTLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, acts, labels, act_lens, label_lens, blank, reduction, fastemit_lambda, clamp):
        certify_inputs(acts, labels, act_lens, label_lens)
        if clamp < 0:
            raise ValueError("`clamp` must be 0.0 or positive float value.")
        B = acts.size(0)
        grads = torch.empty_like(acts, dtype=torch.float32) if acts.requires_grad else None
        scale = 1.0 / B if reduction == "mean" else 1.0  # rnnt_pytorch.py:77-80, folded into the gradient kernel
        costs = ops.rnnt_loss(acts, labels, act_lens, label_lens, blank, grads=grads, fastemit_lambda=fastemit_lambda,
                              clamp=clamp, grad_scale=scale)
        if reduction in ("sum", "mean"):
            costs = costs.sum().unsqueeze_(-1)
            if reduction == "mean":
                costs /= B
        ctx.save_for_backward(grads)
        return costs

    @staticmethod
    def backward(ctx, grad_output):
        (grads,) = ctx.saved_tensors
        if grad_output is not None and grads is not None:
            return grads.mul_(grad_output.view(-1, 1, 1, 1).to(grads)), None, None, None, None, None, None, None
        return (None,) * 8


class RNNTLoss(nn.Module):
    """`RNNTLossNumba(blank=0, reduction='mean', fastemit_lambda=0.0, clamp=-1)`"""

    def __init__(self, blank: int = 0, reduction: str = "mean", fastemit_lambda: float = 0.0, clamp: float = -1):
        super().__init__()
        self.blank = blank
        self.fastemit_lambda = fastemit_lambda
        self.clamp = float(clamp) if clamp > 0 else 0.0
        self.reduction = reduction

    def forward(self, acts, labels, act_lens, label_lens):
        """acts (batch x seqLength x labelLength x outputDim) logits; labels zero-padded [B, U]; lens [B]"""
        if not acts.is_cuda:
            raise RuntimeError("nemo_amd RNNTLoss runs on MI355X only (there is no CPU fallback)")
        if acts.dtype != torch.float32:  # rnnt_pytorch.py:419-423: the numba loss computes in fp32
            acts = acts.float()
        acts = acts.contiguous()
        return _RNNTLossFn.apply(acts, labels.contiguous(), act_lens.contiguous(), label_lens.contiguous(), self.blank,
                                 self.reduction, self.fastemit_lambda, self.clamp)


RNNTLossNumba = RNNTLoss
