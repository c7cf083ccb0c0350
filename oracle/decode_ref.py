"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's greedy CTC decoding + WER.

  * greedy_decode      = GreedyCTCInfer._greedy_decode_logprobs (parts/submodules/ctc_greedy_decoding.py:333-361: per-frame
                         max / arg-max over the first `out_len` frames, score = sum of the max log-probs of non-blank frames)
                         followed by the CTC collapse loop of AbstractCTCDecoding.decode_hypothesis
                         (parts/submodules/ctc_decoding.py:545-575: keep p iff (p != previous or previous == blank) and
                         p != blank).
  * tokens_to_text     = CTCDecoding.decode_tokens_to_str (ctc_decoding.py:1075-1086: ''.join of the vocabulary entries).
  * word_error_rate    = nemo/collections/asr/metrics/wer.py:35-73 with a plain Levenshtein distance in place of the
                         third-party `editdistance.eval` (pinned by the reference's own vectors,
                         tests/collections/asr/test_asr_metrics.py:119-124, in tests/test_oracle_pinning.py).
Nothing under nemo_amd/ imports this file.
"""
from __future__ import annotations

import torch


def greedy_decode(logp: torch.Tensor, lens, blank: int):
    """logp [B,T,C] -> list of (token list, score)"""
    out = []
    for b in range(logp.shape[0]):
        n = int(lens[b]) if lens is not None else logp.shape[1]
        pred = logp[b, :n].float().cpu()
        lp, lab = pred.max(dim=-1)
        score = float(lp[lab != blank].sum()) if n > 0 else 0.0
        toks, previous = [], blank
        for p in lab.tolist():
            if (p != previous or previous == blank) and p != blank:
                toks.append(p)
            previous = p
        out.append((toks, score))
    return out


def tokens_to_text(tokens, vocabulary):
    return "".join(vocabulary[t] for t in tokens)


def levenshtein(a, b) -> int:
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def word_error_rate(hypotheses, references, use_cer=False) -> float:
    if len(hypotheses) != len(references):
        raise ValueError("In word error rate calculation, hypotheses and reference lists must have the same number of "
                         "elements. But I got:{0} and {1} correspondingly".format(len(hypotheses), len(references)))
    scores = words = 0
    for h, r in zip(hypotheses, references):
        h_list, r_list = (list(h), list(r)) if use_cer else (h.split(), r.split())
        words += len(r_list)
        scores += levenshtein(h_list, r_list)
    return 1.0 * scores / words if words != 0 else float("inf")
