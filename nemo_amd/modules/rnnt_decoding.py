"""Greedy transducer decoding and WER for the drop-in FastConformer-Transducer model.

Mirrors the pieces of the reference a training run touches (names, argument meaning, return shapes):
  * `GreedyBatchedRNNTInfer` (parts/submodules/rnnt_greedy_decoding.py:529): `forward(encoder_output [B, D, T], encoded_lengths)`
    -> a 1-tuple holding the list of `Hypothesis` (y_sequence, timestamp, score);
  * `RNNTDecoding.rnnt_decoder_predictions_tensor` (parts/submodules/rnnt_decoding.py:430-520): hypotheses with `.text`;
  * `WER` for transducers (metrics/wer.py:210-356: `update(predictions = encoder output, predictions_lengths, targets,
    targets_lengths)` decodes, then accumulates edit distance / reference words).
The reference's search is a Python loop over frames with a device -> host sync per inner iteration; here the whole batch is ONE
launch (`mi355x_rnnt_greedy_decode`, csrc/rnnt_decode.hip): the encoder projection is a GEMM, a workgroup per utterance runs the
LSTM / joint / arg-max recurrence out of LDS, and only the token ids leave the device -- when text is asked for.

Two documented differences to the reference's search (token ids / time stamps / lengths are bit-identical to it, tests/test_rnnt_decoding.py):
  * `Hypothesis.score` is the sum of the emitted labels' log-probabilities (the reference's CPU behaviour); on CUDA tensors the
    reference sums raw maximum logits because `_joint_step(log_normalize=None)` skips log_softmax there (rnnt_greedy_decoding.py:257-259);
  * `max_symbols_per_step=None` is unbounded per frame in the reference; here an utterance stops once 4 * T labels are out
    (a model that never emits blank cannot hang the device; `out_len == 4 * T` marks the cut).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch

from .. import ops
from .ctc_decoding import _levenshtein


@dataclass
class Hypothesis:  # parts/utils/rnnt_utils.py:35-110 (the fields greedy decoding fills)
    score: float
    y_sequence: torch.Tensor
    timestamp: List[int] = field(default_factory=list)
    text: Optional[str] = None
    dec_state: Optional[tuple] = None
    length: int = 0


class GreedyBatchedRNNTInfer:
    def __init__(self, decoder_model, joint_model, blank_index: int, max_symbols_per_step: Optional[int] = None,
                 preserve_alignments: bool = False, preserve_frame_confidence: bool = False, **unused):
        if preserve_alignments or preserve_frame_confidence:
            raise NotImplementedError("alignments / frame confidences are not produced by the on-device search")
        if max_symbols_per_step is not None and max_symbols_per_step <= 0:
            raise ValueError(f"Expected max_symbols_per_step > 0 (or None), got {max_symbols_per_step}")  # rnnt_greedy_decoding.py:187
        if getattr(decoder_model, "pred_rnn_layers", 1) != 1:
            raise NotImplementedError("on-device greedy search: one LSTM layer in the prediction network (the recipe's pred_rnn_layers)")
        self.decoder, self.joint = decoder_model, joint_model
        self._blank_index = int(blank_index)
        self.max_symbols = max_symbols_per_step

    @torch.no_grad()
    def decode_ids(self, encoder_output: torch.Tensor, encoded_lengths: torch.Tensor, with_state: bool = False):
        """encoder_output [B, D, T] (device) -> (tokens i32 [B, N] -1 padded, frame indices, lengths i32 [B], scores f32 [B])"""
        if encoder_output.dim() != 3:
            raise ValueError(f"`encoder_output` must be [B, D, T]; got shape {tuple(encoder_output.shape)}")
        dec, jnt = self.decoder, self.joint
        dev = encoder_output.device
        B, D, T = encoder_output.shape
        cdt = jnt._cdt()
        lstm = dec.prediction["dec_rnn"].lstm
        emb = dec.prediction["embed"].weight
        out = jnt.joint_net[-1]
        xe32 = encoder_output.transpose(1, 2).contiguous().view(B * T, D).float()
        J = jnt.joint_hidden
        if cdt == torch.bfloat16:   # the GEMM operand images the training step keeps up to date
            Wj, Wd = jnt._plan(cdt, dev), dec._plan(cdt, dev)
            xe = torch.empty(B * T, D, dtype=cdt, device=dev)
            ops.drop_scale_cast(xe32, xe, B * T * D, 1.0)
            f = torch.empty(B * T, J, dtype=cdt, device=dev)
            ops.gemm(xe, Wj["enc.w"], f, B * T, J, D, D, Wj.pitch("enc.w"), J, bias=jnt.enc.bias)
            w = (Wd["l0.wih"], Wd.pitch("l0.wih"), Wd["l0.whh"], Wd.pitch("l0.whh"), Wj["pred.w"], Wj.pitch("pred.w"), Wj["out.w"],
                 Wj.pitch("out.w"))
        else:
            f = torch.empty(B * T, J, dtype=torch.float32, device=dev)
            ops.gemm(xe32, jnt.enc.weight, f, B * T, J, D, D, D, J, bias=jnt.enc.bias)
            H = dec.pred_hidden
            w = (lstm.weight_ih_l0, H, lstm.weight_hh_l0, H, jnt.pred.weight, H, out.weight, J)
        lens = encoded_lengths.to(device=dev, dtype=torch.int64).contiguous()
        res = ops.rnnt_greedy_decode(f.view(B, T, J), lens, emb, w[0], w[1], w[2], w[3], lstm.bias_ih_l0, lstm.bias_hh_l0, w[4], w[5],
                                     jnt.pred.bias, w[6], w[7], out.bias, self._blank_index, self.max_symbols or 0,
                                     with_state=with_state)
        return res

    def forward(self, encoder_output: torch.Tensor, encoded_lengths: torch.Tensor, partial_hypotheses=None):
        if partial_hypotheses is not None:
            raise NotImplementedError("`partial_hypotheses` support is not supported")  # as the frame-looping reference path (:816)
        tokens, times, out_len, score, (h, c) = self.decode_ids(encoder_output, encoded_lengths, with_state=True)
        tokens, times, out_len, score = tokens.cpu(), times.cpu(), out_len.cpu(), score.cpu()   # the only D2H copies
        hyps = []
        for b in range(tokens.shape[0]):
            n = int(out_len[b])
            hyps.append(Hypothesis(score=float(score[b]), y_sequence=tokens[b, :n].to(torch.long), timestamp=times[b, :n].tolist(),
                                   dec_state=(h[b], c[b]), length=int(encoded_lengths[b])))
        return (hyps,)

    __call__ = forward


class RNNTDecoding:
    """`strategy: greedy_batch` of AbstractRNNTDecoding (rnnt_decoding.py:216-330) for a character / word-piece vocabulary;
    blank id = len(vocabulary) (rnnt_decoding.py:1170)."""

    def __init__(self, decoder, joint, vocabulary: Optional[Sequence[str]] = None, max_symbols: Optional[int] = 10,
                 tokenizer=None):
        self.vocabulary = list(vocabulary) if vocabulary is not None else None
        self.tokenizer = tokenizer
        self.blank_id = decoder.blank_idx
        self.decoding = GreedyBatchedRNNTInfer(decoder, joint, self.blank_id, max_symbols_per_step=max_symbols)

    def ids_to_text(self, ids: Sequence[int]) -> str:
        ids = [int(i) for i in ids if int(i) != self.blank_id]
        if self.tokenizer is not None:
            return self.tokenizer.ids_to_text(ids)
        if self.vocabulary is None:
            raise ValueError("no vocabulary: text is not available")
        text = "".join(self.vocabulary[i] for i in ids)
        return text.replace("▁", " ").strip() if "▁" in text else text

    def rnnt_decoder_predictions_tensor(self, encoder_output: torch.Tensor, encoded_lengths: torch.Tensor,
                                        return_hypotheses: bool = False):
        hyps = self.decoding(encoder_output=encoder_output, encoded_lengths=encoded_lengths)[0]
        for h in hyps:
            h.text = self.ids_to_text(h.y_sequence.tolist())
        return hyps


class RNNTWER:
    """metrics/wer.py:210-356 with a transducer decoding object: `predictions` = the ENCODER output [B, D, T]"""

    def __init__(self, decoding: RNNTDecoding, use_cer: bool = False):
        self.decoding, self.use_cer = decoding, use_cer
        self.scores = 0
        self.words = 0
        self._to_sync = True

    def update(self, predictions: torch.Tensor, predictions_lengths, targets: torch.Tensor, targets_lengths):
        hyps = [h.text for h in self.decoding.rnnt_decoder_predictions_tensor(predictions, predictions_lengths)]
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
