"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the CTC loss / gradient used on the hot path.

Reference boundary: nemo/collections/asr/losses/ctc.py:25-82 (`CTCLoss(nn.CTCLoss)`): the arithmetic is
third-party `torch.nn.functional.ctc_loss` (torch is unpinned in requirements/requirements.txt:14; the
build container has 2.10.0+rocm7.0).  This file restates the published algorithm (Graves et al. 2006,
log-space alpha/beta over the blank-extended label sequence, `zero_infinity`) independently of torch, in
float64, so it can arbitrate between the HIP kernel and torch.

Pinned by: the warp-ctc known-answer vectors the reference keeps in
tests/collections/asr/k2/test_ctc.py:85-120,124-187,209-284 (committed as
tests/golden/ctc_known_answers.json by oracle/make_golden.py) -- see tests/test_oracle_pinning.py.
"""
from __future__ import annotations

import numpy as np

NEG_INF = -np.inf


def _logaddexp3(a, b, c):
    m = max(a, b, c)
    if m == NEG_INF:
        return NEG_INF
    return m + np.log(np.exp(a - m) + np.exp(b - m) + np.exp(c - m))


def ctc_alpha_beta(logp: np.ndarray, target: np.ndarray, blank: int):
    """logp [T, C] log-probabilities, target [U] -> (nll, alpha [T,S], beta [T,S]) with S = 2U+1."""
    T, C = logp.shape
    U = len(target)
    S = 2 * U + 1
    ext = np.full(S, blank, dtype=np.int64)
    ext[1::2] = target
    alpha = np.full((T, S), NEG_INF)
    beta = np.full((T, S), NEG_INF)
    if T == 0:
        return (0.0 if U == 0 else np.inf), alpha, beta
    alpha[0, 0] = logp[0, blank]
    if S > 1:
        alpha[0, 1] = logp[0, ext[1]]
    for t in range(1, T):
        for s in range(S):
            a = alpha[t - 1, s]
            b = alpha[t - 1, s - 1] if s >= 1 else NEG_INF
            c = alpha[t - 1, s - 2] if (s >= 2 and ext[s] != blank and ext[s] != ext[s - 2]) else NEG_INF
            v = _logaddexp3(a, b, c)
            alpha[t, s] = v + logp[t, ext[s]] if v != NEG_INF else NEG_INF
    beta[T - 1, S - 1] = logp[T - 1, blank]
    if S > 1:
        beta[T - 1, S - 2] = logp[T - 1, ext[S - 2]]
    for t in range(T - 2, -1, -1):
        for s in range(S):
            a = beta[t + 1, s]
            b = beta[t + 1, s + 1] if s + 1 < S else NEG_INF
            c = beta[t + 1, s + 2] if (s + 2 < S and ext[s] != blank and ext[s] != ext[s + 2]) else NEG_INF
            v = _logaddexp3(a, b, c)
            beta[t, s] = v + logp[t, ext[s]] if v != NEG_INF else NEG_INF
    ll = np.logaddexp(alpha[T - 1, S - 1], alpha[T - 1, S - 2] if S > 1 else NEG_INF)
    return -ll, alpha, beta


def ctc_loss_and_grad(logp: np.ndarray, targets, in_len, tgt_len, blank: int, zero_infinity: bool = True):
    """logp [B, T, C] (log-softmax output), targets [B, Umax] -> (nll [B], dL/dlogp [B, T, C]) where L = sum_b nll_b.

    grad wrt log-probs (what autograd gives for F.ctc_loss): -exp(alpha+beta - logp - ll) scattered per class.
    Frames t >= in_len[b] get zero gradient."""
    logp = np.asarray(logp, dtype=np.float64)
    B, Tmax, C = logp.shape
    nll = np.zeros(B)
    grad = np.zeros_like(logp)
    for b in range(B):
        T, U = int(in_len[b]), int(tgt_len[b])
        tgt = np.asarray(targets[b][:U], dtype=np.int64)
        loss, alpha, beta = ctc_alpha_beta(logp[b, :T], tgt, blank)
        if not np.isfinite(loss):
            nll[b] = 0.0 if zero_infinity else np.inf
            continue
        nll[b] = loss
        ext = np.full(2 * U + 1, blank, dtype=np.int64)
        ext[1::2] = tgt
        ab = alpha + beta  # includes logp[t, ext[s]] twice
        for t in range(T):
            for s in range(2 * U + 1):
                if ab[t, s] != NEG_INF:
                    c = ext[s]
                    grad[b, t, c] -= np.exp(ab[t, s] - logp[b, t, c] + loss)
    return nll, grad


def ctc_loss_and_grad_wrt_logits(acts: np.ndarray, targets, in_len, tgt_len, blank: int):
    """acts [B,T,C] raw activations -> (nll [B], d(sum nll)/d acts), i.e. the harness of
    tests/collections/asr/k2/test_ctc.py:26-59 (log_softmax -> loss -> sum -> backward)."""
    acts = np.asarray(acts, dtype=np.float64)
    m = acts.max(-1, keepdims=True)
    logp = acts - m - np.log(np.exp(acts - m).sum(-1, keepdims=True))
    nll, g = ctc_loss_and_grad(logp, targets, in_len, tgt_len, blank)
    # log-softmax backward: dx = g - softmax * sum(g)
    gx = g - np.exp(logp) * g.sum(-1, keepdims=True)
    return nll, gx
