"""Greedy CTC decoding and WER for the drop-in model.

The reference decodes on the host: `GreedyCTCInfer` copies the log-probabilities to the CPU utterance by utterance and
folds them in Python (parts/submodules/ctc_greedy_decoding.py:333-361, ctc_decoding.py:545-575), a D2H sync of
[B, T, V+1] floats (8 MB at the headline shape) on every logging step.  Here arg-max, score, repeat folding and blank
removal are one kernel launch (`mi355x_ctc_greedy_decode`); only the folded token ids (<= 64 KB) ever leave the device,
and only when text is asked for.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from .. import ops


class GreedyCTCDecoder:
    """blank_id = len(vocabulary) as in CTCDecoding.__init__ (ctc_decoding.py:1044); `vocabulary` maps ids to strings
    (characters, or word pieces whose leading U+2581 marks a word start -- the SentencePiece convention)."""

    def __init__(self, vocabulary: Optional[Sequence[str]] = None, blank_id: Optional[int] = None):
        if vocabulary is None and blank_id is None:
            raise ValueError("either a vocabulary or a blank_id is required")
        self.vocabulary = list(vocabulary) if vocabulary is not None else None
        self.blank_id = len(self.vocabulary) if blank_id is None else int(blank_id)

    @torch.no_grad()
    def decode_ids(self, log_probs: torch.Tensor, lengths: Optional[torch.Tensor] = None):
        """log_probs [B, T, V+1] (device) -> (tokens i32 [B,T] folded / blank-free / -1 padded, lengths i32 [B], score [B])"""
        if log_probs.dim() != 3:
            raise ValueError(f"`decoder_output` must be a tensor of shape [B, T, V] (log probs, float). Provided shape = "
                             f"{tuple(log_probs.shape)}")
        lp = log_probs.to(torch.float32).contiguous()
        lens = lengths.to(torch.int64).contiguous() if lengths is not None else None
        return ops.ctc_greedy_decode(lp, lens, self.blank_id)

    def ids_to_text(self, ids: Sequence[int]) -> str:
        if self.vocabulary is None:
            raise ValueError("no vocabulary: text is not available")
        text = "".join(self.vocabulary[i] for i in ids)             # decode_tokens_to_str, ctc_decoding.py:1075-1086
        return text.replace("▁", " ").strip() if "▁" in text else text

    def __call__(self, log_probs, lengths=None) -> List[str]:
        tokens, out_len, _ = self.decode_ids(log_probs, lengths)
        tokens, out_len = tokens.cpu(), out_len.cpu()                  # the only D2H copy: folded ids
        return [self.ids_to_text(tokens[b, : int(out_len[b])].tolist()) for b in range(tokens.shape[0])]


def _levenshtein(a, b) -> int:
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def word_error_rate(hypotheses: List[str], references: List[str], use_cer: bool = False) -> float:
    """nemo/collections/asr/metrics/wer.py:35-73"""
    if len(hypotheses) != len(references):
        raise ValueError("In word error rate calculation, hypotheses and reference lists must have the same number of "
                         "elements. But I got:{0} and {1} correspondingly".format(len(hypotheses), len(references)))
    scores = words = 0
    for h, r in zip(hypotheses, references):
        h_list, r_list = (list(h), list(r)) if use_cer else (h.split(), r.split())
        words += len(r_list)
        scores += _levenshtein(h_list, r_list)
    return 1.0 * scores / words if words != 0 else float("inf")


class WER:
    """metrics/wer.py:210-356: accumulates edit distance and reference word counts over batches; `compute()` returns
    (wer, scores, words) like the torchmetrics object of the reference."""

    def __init__(self, decoding: GreedyCTCDecoder, use_cer: bool = False):
        self.decoding, self.use_cer = decoding, use_cer
        self.scores = 0
        self.words = 0

    def update(self, predictions: torch.Tensor, predictions_lengths, targets: torch.Tensor, targets_lengths):
        hyps = self.decoding(predictions, predictions_lengths)
        tg, tl = targets.cpu(), targets_lengths.cpu()
        refs = [self.decoding.ids_to_text(tg[b, : int(tl[b])].tolist()) for b in range(tg.shape[0])]
        for h, r in zip(hyps, refs):
            h_list, r_list = (list(h), list(r)) if self.use_cer else (h.split(), r.split())
            self.words += len(r_list)
            self.scores += _levenshtein(h_list, r_list)

    def compute(self):
        wer = self.scores / self.words if self.words else float("inf")
        return wer, self.scores, self.words

    def reset(self):
        self.scores = self.words = 0
