"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's RNN-Transducer loss (never imported by nemo_amd/).

Follows, in float64 torch on the host:
  * the lattice recursions of `nemo/collections/asr/parts/numba/rnnt_loss/rnnt_numpy.py:131-187` (`forward_pass`,
    `backward_pass`: alphas[t,u] = logaddexp(alphas[t-1,u] + lp[t-1,u,blank], alphas[t,u-1] + lp[t,u-1,label[u-1]]), ...);
  * the fused gradient w.r.t. the LOGITS of the GPU path, `utils/cuda_utils/gpu_rnnt_kernel.py:355-396` (softmax Jacobian
    folded in, FastEmit term :364-376, label term scaled by log1p(lambda) :387-388, clamp :392-396);
  * costs = -(1 + fastemit_lambda) * loglike (`rnnt_helper.py:107-116`), reductions and the 1/B gradient scale of 'mean'
    (`rnnt_pytorch.py:75-80`).
Pinned by tests/test_oracle_pinning.py against the known-answer costs and gradients of the reference's own tests
(`tests/collections/asr/numba/rnnt_loss/test_rnnt_pytorch.py:82-128,190-310,358-402`, extracted by oracle/make_golden.py
into tests/golden/rnnt_known_answers.json) and against autograd through an independent log-softmax + DP."""
from __future__ import annotations

import math

import torch


def _lattice(lp, labels, T, U1, blank):
    """lp [T,U1,V1] log-probs (f64) of ONE utterance, labels list[int] (len U1-1) -> alphas, betas [T,U1], loglike"""
    ninf = -math.inf
    a = torch.full((T, U1), ninf, dtype=torch.float64)
    b = torch.full((T, U1), ninf, dtype=torch.float64)
    a[0, 0] = 0.0
    for t in range(1, T):
        a[t, 0] = a[t - 1, 0] + lp[t - 1, 0, blank]
    for u in range(1, U1):
        a[0, u] = a[0, u - 1] + lp[0, u - 1, labels[u - 1]]
    for t in range(1, T):
        for u in range(1, U1):
            a[t, u] = torch.logaddexp(a[t, u - 1] + lp[t, u - 1, labels[u - 1]], a[t - 1, u] + lp[t - 1, u, blank])
    b[T - 1, U1 - 1] = lp[T - 1, U1 - 1, blank]
    for t in reversed(range(T - 1)):
        b[t, U1 - 1] = b[t + 1, U1 - 1] + lp[t, U1 - 1, blank]
    for u in reversed(range(U1 - 1)):
        b[T - 1, u] = b[T - 1, u + 1] + lp[T - 1, u, labels[u]]
    for t in reversed(range(T - 1)):
        for u in reversed(range(U1 - 1)):
            b[t, u] = torch.logaddexp(b[t, u + 1] + lp[t, u, labels[u]], b[t + 1, u] + lp[t, u, blank])
    return a, b, a[T - 1, U1 - 1] + lp[T - 1, U1 - 1, blank]


def rnnt_loss_and_grad(acts, labels, act_lens, label_lens, blank=0, fastemit_lambda=0.0, clamp=0.0, reduction="sum"):
    """acts [B,T,U1,V1] logits; returns (costs: [B] for 'none' else [1], grads [B,T,U1,V1]) in float64"""
    acts = acts.detach().to(torch.float64)
    B, T, U1, V1 = acts.shape
    costs = torch.zeros(B, dtype=torch.float64)
    grads = torch.zeros_like(acts)
    for i in range(B):
        Tb, Ub = int(act_lens[i]), int(label_lens[i]) + 1
        lab = [int(v) for v in labels[i, : Ub - 1]]
        x = acts[i, :Tb, :Ub]
        lp = torch.log_softmax(x, dim=-1)
        a, b, ll = _lattice(lp, lab, Tb, Ub, blank)
        costs[i] = -ll * (1.0 + fastemit_lambda)
        g = torch.exp(a[:, :, None] + b[:, :, None] + lp - ll)
        for t in range(Tb):
            for u in range(Ub):
                if fastemit_lambda > 0.0 and u < Ub - 1:
                    g[t, u] += fastemit_lambda * torch.exp(a[t, u] + lp[t, u, lab[u]] + b[t, u + 1] + lp[t, u] - ll)
                if t == Tb - 1 and u == Ub - 1:
                    g[t, u, blank] -= torch.exp(a[t, u] + lp[t, u, blank] - ll)
                if t < Tb - 1:
                    g[t, u, blank] -= torch.exp(a[t, u] + lp[t, u, blank] - ll + b[t + 1, u])
                if u < Ub - 1:
                    g[t, u, lab[u]] -= torch.exp(math.log1p(fastemit_lambda) + a[t, u] + lp[t, u, lab[u]] - ll + b[t, u + 1])
        if clamp > 0.0:
            g = g.clamp(-clamp, clamp)
        grads[i, :Tb, :Ub] = g
    if reduction in ("sum", "mean"):
        costs = costs.sum().unsqueeze(-1)
        if reduction == "mean":
            costs = costs / B
            grads = grads / B
    return costs, grads


def rnnt_nll_autograd(acts, labels, act_lens, label_lens, blank=0):
    """independent check of the closed form: -log P(y|x) by the alpha recursion alone, differentiated by autograd"""
    acts = acts.detach().to(torch.float64).requires_grad_(True)
    total = 0.0
    for i in range(acts.shape[0]):
        Tb, Ub = int(act_lens[i]), int(label_lens[i]) + 1
        lab = [int(v) for v in labels[i, : Ub - 1]]
        lp = torch.log_softmax(acts[i, :Tb, :Ub], dim=-1)
        prev = None
        for t in range(Tb):
            row = []
            for u in range(Ub):
                if t == 0 and u == 0:
                    v = lp.new_zeros(())
                else:
                    terms = []
                    if t > 0:
                        terms.append(prev[u] + lp[t - 1, u, blank])
                    if u > 0:
                        terms.append(row[u - 1] + lp[t, u - 1, lab[u - 1]])
                    v = terms[0] if len(terms) == 1 else torch.logaddexp(terms[0], terms[1])
                row.append(v)
            prev = row
        total = total - (prev[Ub - 1] + lp[Tb - 1, Ub - 1, blank])
    total.backward()
    return total.detach(), acts.grad
