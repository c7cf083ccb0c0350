"""Drop-in for `nemo.collections.asr.models.EncDecRNNTModel` / `EncDecRNNTBPEModel` (models/rnnt_models.py) restricted to the
training hot path of FastConformer-Transducer (BASELINE.json configs[3]): config-driven construction of preprocessor /
encoder / decoder (prediction network) / joint / loss (rnnt_models.py:50-120), `forward` (:630-690: audio -> encoder
output), `training_step` (:692-760: decoder -> joint, fused with the loss when `joint.fuse_loss_wer`), and the optimizer /
data-parallel machinery shared with the CTC model (`fit_step`, bucketed gradient exchange over the flat buffers of encoder,
prediction network and joint), greedy batched decoding + WER on the device (`decoding`, `wer`, `validation_pass`, `transcribe`:
rnnt_models.py:92-110, 245-330, 790-850; modules/rnnt_decoding.py).  Beam search is outside this path."""
from __future__ import annotations

import copy
import os
from typing import Any, Dict

import torch

from ..modules import RNNTLoss
from .ctc_models import EncDecCTCModel, _build as _build_ctc

_DEFAULT_TARGETS = {"decoder": "nemo.collections.asr.modules.RNNTDecoder", "joint": "nemo.collections.asr.modules.RNNTJoint"}


def _build(section, cfg):
    from ..core import Serialization
    cfg = dict(cfg)
    cfg.setdefault("_target_", _DEFAULT_TARGETS[section])
    return Serialization.from_config_dict(cfg)


class EncDecRNNTModel(EncDecCTCModel):
    def __init__(self, cfg: Dict[str, Any], trainer=None):
        torch.nn.Module.__init__(self)
        cfg = copy.deepcopy(dict(cfg))
        self._cfg = cfg
        self.trainer = trainer
        self.preprocessor = _build_ctc("preprocessor", cfg["preprocessor"])
        self.encoder = _build_ctc("encoder", cfg["encoder"])
        # rnnt_models.py:60-90: vocabulary size and hidden sizes are injected into the decoder / joint sections
        vocab = cfg.get("labels")
        self.tokenizer = None
        tok_cfg = cfg.get("tokenizer")
        if vocab is None and tok_cfg and (tok_cfg.get("dir") or tok_cfg.get("model_path")):
            # EncDecRNNTBPEModel (rnnt_bpe_models.py:40-90): the SentencePiece pieces in id order are the vocabulary
            from ..data import SentencePieceTokenizer
            if str(tok_cfg.get("type", "bpe")).lower() != "bpe":
                raise NotImplementedError("WordPiece (BERT) tokenizers: the transducer recipes use SentencePiece `bpe`")
            model_path = tok_cfg.get("model_path") or os.path.join(tok_cfg["dir"], "tokenizer.model")
            self.tokenizer = SentencePieceTokenizer(model_path)
            vocab = self.tokenizer.vocab
            cfg["tokenizer"] = dict(tok_cfg, model_path=model_path, type="bpe")
        n_cls = len(vocab) if vocab is not None else int(cfg["joint"].get("num_classes", cfg["decoder"].get("vocab_size", -1)))
        if n_cls < 1:
            raise ValueError("the vocabulary size must be given (`labels`, `joint.num_classes` or `decoder.vocab_size`)")
        dec = dict(cfg["decoder"]); dec["vocab_size"] = n_cls
        jnt = dict(cfg["joint"]); jnt["num_classes"] = n_cls
        jnt["jointnet"] = dict(jnt["jointnet"])
        jnt["jointnet"].setdefault("encoder_hidden", cfg.get("model_defaults", {}).get("enc_hidden", self.encoder._feat_out))
        jnt["jointnet"].setdefault("pred_hidden", dec["prednet"]["pred_hidden"])
        self.decoder = _build("decoder", dec)
        self.joint = _build("joint", jnt)
        lc = dict(cfg.get("loss") or {})
        if lc.get("loss_name", "default") not in ("default", "warprnnt_numba"):
            # losses/rnnt.py:41-100 RNNT_LOSS_RESOLVER: tdt / multiblank / pytorch / graph losses are different objectives
            raise NotImplementedError(f"transducer loss '{lc.get('loss_name')}' (implemented: default = warprnnt_numba semantics)")
        if cfg.get("aux_ctc"):
            raise NotImplementedError("aux_ctc (EncDecHybridRNNTCTCModel's auxiliary CTC head) is not part of EncDecRNNTModel")
        self._check_interctc(cfg.get("interctc"))
        # skip_nan_grad (models/asr_model.py:147-174): handled by the shared fit_step / on_after_backward (EncDecCTCModel)
        self._skip_nan_grad = bool(cfg.get("skip_nan_grad"))
        self.skipped_steps = 0
        kw = dict(lc.get("warprnnt_numba_kwargs") or {})
        self.loss = RNNTLoss(blank=n_cls, reduction=cfg.get("rnnt_reduction", "mean_batch"),
                             fastemit_lambda=kw.get("fastemit_lambda", 0.0), clamp=kw.get("clamp", -1.0))
        if self.joint.fuse_loss_wer:
            self.joint.set_loss(self.loss)
        sa = cfg.get("spec_augment")
        self.spec_augmentation = _build_ctc("spec_augment", sa) if sa else None
        self._optimizer = self._scheduler = self._syncs = None
        self._wer = None
        self._decoding = None
        self.validation_step_outputs, self.test_step_outputs = [], []
        self.optimizer_in_backward = False
        self.global_step = 0
        self.pred_side_stream = os.environ.get("MI355X_PRED_STREAM", "1") != "0"
        self._pred_stream = None

    def trainable_modules(self):
        return [self.encoder, self.decoder, self.joint]

    def _after_backward(self):
        # the prediction network's backward ran on its own stream (autograd replays a node on the stream of its forward) and
        # wrote its gradients straight into the flat buffer, not through AccumulateGrad: order the optimizer behind it explicitly
        if self._pred_stream is not None:
            torch.cuda.current_stream(self._pred_stream.device).wait_stream(self._pred_stream)

    def _artifacts(self):
        return {"tokenizer.model_path": self._cfg["tokenizer"]["model_path"]} if self.tokenizer is not None else {}

    @property
    def decoding(self):
        """greedy batched transducer decoding over the model's vocabulary (rnnt_models.py:92-100: RNNTDecoding with the recipe's
        `decoding: {strategy: greedy_batch, greedy: {max_symbols: 10}}`), on the device (modules/rnnt_decoding.py)"""
        if self._decoding is None:
            vocab = self._cfg.get("labels")
            if vocab is None and self.tokenizer is None:
                return None
            from ..modules import RNNTDecoding
            dcfg = dict(self._cfg.get("decoding") or {})
            strategy = dcfg.get("strategy", "greedy_batch")
            if strategy not in ("greedy", "greedy_batch"):
                raise NotImplementedError(f"transducer decoding strategy '{strategy}' (implemented: greedy, greedy_batch)")
            ms = dict(dcfg.get("greedy") or {}).get("max_symbols", 10)
            self._decoding = RNNTDecoding(self.decoder, self.joint, vocabulary=list(vocab) if vocab is not None else None,
                                          max_symbols=ms, tokenizer=self.tokenizer if vocab is None else None)
        return self._decoding

    @property
    def wer(self):
        """rnnt_models.py:101-110: WER over the greedy hypotheses (`predictions` = encoder output)"""
        if self._wer is None and self.decoding is not None:
            from ..modules import RNNTWER
            self._wer = RNNTWER(self.decoding, use_cer=bool(self._cfg.get("use_cer", False)))
            if self.joint.fuse_loss_wer:
                self.joint.set_wer(self._wer)
        return self._wer

    # ------------------------------------------------------------------ forward (rnnt_models.py:630-690)
    def forward(self, input_signal=None, input_signal_length=None, processed_signal=None, processed_signal_length=None):
        has_input_signal = input_signal is not None and input_signal_length is not None
        has_processed_signal = processed_signal is not None and processed_signal_length is not None
        if (has_input_signal ^ has_processed_signal) is False:
            raise ValueError(f"{self} Arguments ``input_signal`` and ``input_signal_length`` are mutually exclusive "
                             " with ``processed_signal`` and ``processed_signal_len`` arguments.")
        if not has_processed_signal:
            processed_signal, processed_signal_length = self.preprocessor(input_signal=input_signal, length=input_signal_length)
        if self.spec_augmentation is not None and self.training:
            processed_signal = self.spec_augmentation(input_spec=processed_signal, length=processed_signal_length)
        return self.encoder(audio_signal=processed_signal, length=processed_signal_length)

    # ------------------------------------------------------------------ training_step (rnnt_models.py:692-760)
    def training_step(self, batch, batch_nb=0):
        signal, signal_len, transcript, transcript_len = batch
        # rnnt_models.py:720-724: the training WER is computed every `log_every_n_steps` steps (greedy decoding of the batch)
        n_log = self._log_every_n_steps()
        compute_wer = bool(n_log) and (batch_nb + 1) % n_log == 0 and self.wer is not None and torch.is_grad_enabled()
        # The prediction network is 2 x (U+1) tiny dependent launches (recurrent GEMM + cell kernel per step): latency-bound and
        # independent of the encoder until the joint.  It runs on its own stream next to the encoder forward; autograd replays
        # a node's backward on the stream of its forward, so BPTT overlaps the encoder backward the same way.
        if self.pred_side_stream and signal.is_cuda:
            cur = torch.cuda.current_stream(signal.device)
            if self._pred_stream is None or self._pred_stream.device != signal.device:
                from ..streams import private_stream
                self._pred_stream = private_stream(signal.device)
            self._pred_stream.wait_stream(cur)
            with torch.cuda.stream(self._pred_stream):
                decoder, target_length, _ = self.decoder(targets=transcript, target_length=transcript_len)
            encoded, encoded_len = self.forward(input_signal=signal, input_signal_length=signal_len)
            cur.wait_stream(self._pred_stream)
            decoder.record_stream(cur)
            target_length.record_stream(cur)
        else:
            encoded, encoded_len = self.forward(input_signal=signal, input_signal_length=signal_len)
            decoder, target_length, _ = self.decoder(targets=transcript, target_length=transcript_len)
        loss_value, wer, _, _ = self._loss_and_wer(encoded, encoded_len, decoder, target_length, transcript, transcript_len, compute_wer)
        logs = {"train_loss": loss_value.detach(), "global_step": self.global_step}
        if self._scheduler is not None:
            logs["learning_rate"] = self._scheduler.get_last_lr()
        if wer is not None:
            logs["training_batch_wer"] = wer
        return {"loss": loss_value, "log": logs}

    def _loss_and_wer(self, encoded, encoded_len, decoder, target_length, transcript, transcript_len, compute_wer):
        """joint + loss (+ WER of the greedy hypotheses): rnnt_models.py:725-760 / :815-850 -> (loss, wer, wer_num, wer_denom)"""
        if not self.joint.fuse_loss_wer:
            joint = self.joint(encoder_outputs=encoded, decoder_outputs=decoder)
            # losses/rnnt.py:446-484: a batch padded beyond its longest utterance / transcript is narrowed before the loss
            # (the loss's input check insists on T = max length, U = max target length + 1); two host reads, as in the reference
            max_t, max_u = int(encoded_len.max()), int(target_length.max())
            if joint.shape[1] != max_t or joint.shape[2] != max_u + 1:
                joint = joint[:, :max_t, :max_u + 1].contiguous()
            tr = transcript[:, :max_u] if transcript.shape[1] != max_u else transcript
            loss_value = self._reduce(self.loss(joint, tr.clamp(max=self.loss.blank - 1).contiguous(),
                                                encoded_len.to(torch.int64), target_length.to(torch.int64)), target_length)
            wer = num = denom = None
            if compute_wer:   # rnnt_models.py:742-748 (un-fused joint: the metric is updated after the loss)
                self.wer.update(predictions=encoded.detach(), predictions_lengths=encoded_len, targets=transcript,
                                targets_lengths=transcript_len)
                wer, num, denom = self.wer.compute()
                self.wer.reset()
            return loss_value, wer, num, denom
        return self.joint(encoder_outputs=encoded, decoder_outputs=decoder, encoder_lengths=encoded_len, transcripts=transcript,
                          transcript_lengths=transcript_len, compute_wer=compute_wer)

    def _reduce(self, losses, target_lengths):
        red = self.loss.reduction  # losses/rnnt.py:333-420 (RNNTLoss.reduce)
        if red == "mean_batch":
            return losses.mean()
        if red == "mean":
            return torch.div(losses, target_lengths.clamp(min=1)).mean()
        if red == "sum":
            return losses.sum()
        if red == "mean_volume":
            return losses.sum() / target_lengths.sum()
        return losses

    @torch.no_grad()
    def validation_pass(self, batch, batch_idx=0, dataloader_idx=0):
        """rnnt_models.py:790-850: the loss and the WER numerator / denominator of the batch (greedy hypotheses); with the fused
        joint both come out of its sub-batch loop"""
        signal, signal_len, transcript, transcript_len = batch[:4]
        encoded, encoded_len = self.forward(input_signal=signal, input_signal_length=signal_len)
        decoder, target_length, _ = self.decoder(targets=transcript, target_length=transcript_len)
        loss, wer, num, denom = self._loss_and_wer(encoded, encoded_len, decoder, target_length, transcript, transcript_len,
                                                   self.wer is not None)
        metrics = {"val_loss": loss.detach()}
        if wer is not None:
            metrics.update({"val_wer_num": num, "val_wer_denom": denom, "val_wer": wer})
        return metrics

    @torch.no_grad()
    def predict_step(self, batch, batch_idx=0, dataloader_idx=0):
        signal, signal_len, _, _, sample_id = batch
        encoded, encoded_len = self.forward(input_signal=signal, input_signal_length=signal_len)
        texts = [h.text for h in self.decoding.rnnt_decoder_predictions_tensor(encoded, encoded_len)]
        if isinstance(sample_id, torch.Tensor):
            sample_id = sample_id.cpu().numpy()
        return list(zip(sample_id, texts))

    @torch.no_grad()
    def transcribe(self, audio, batch_size: int = 4, return_hypotheses: bool = False, num_workers: int = 0,
                   channel_selector=None, verbose: bool = False):
        """`ASRTranscriptionMixin.transcribe` for the greedy transducer path (rnnt_models.py:245-330): one string per input, or the
        Hypothesis objects (text, y_sequence, timestamp, score) with return_hypotheses; the order of the inputs is kept"""
        from ..data import load_audio
        if self.decoding is None:
            raise RuntimeError("transcribe() needs a vocabulary (`labels` or a tokenizer)")
        if isinstance(audio, (str, bytes)) or not hasattr(audio, "__len__"):
            audio = [audio]
        sr = self._cfg.get("sample_rate", 16000)
        device = next(self.parameters()).device
        was_training = self.training
        feat = self.preprocessor.featurizer
        dither, pad_to = feat.dither, feat.pad_to
        self.eval()
        feat.dither, feat.pad_to = 0.0, 0
        out = []
        try:
            for i in range(0, len(audio), batch_size):
                waves = []
                for a in audio[i:i + batch_size]:
                    if isinstance(a, str):
                        a = load_audio(a, sr, channel_selector=channel_selector)
                    waves.append(torch.as_tensor(a, dtype=torch.float32).reshape(-1))
                lens = torch.tensor([w.numel() for w in waves], dtype=torch.int64)
                sig = torch.zeros(len(waves), int(lens.max()), dtype=torch.float32)
                for r, w in enumerate(waves):
                    sig[r, : w.numel()] = w
                encoded, enc_len = self.forward(input_signal=sig.to(device), input_signal_length=lens.to(device))
                hyps = self.decoding.rnnt_decoder_predictions_tensor(encoded, enc_len)
                out.extend(hyps if return_hypotheses else [h.text for h in hyps])
        finally:
            feat.dither, feat.pad_to = dither, pad_to
            self.train(was_training)
        return out


def fastconformer_transducer_config(size: str = "large", vocab_size: int = 1024, spec_augment: bool = False,
                                    **encoder_overrides) -> Dict[str, Any]:
    """model section of examples/asr/conf/fastconformer/fast-conformer_transducer_bpe.yaml (sizes of its table: Large = 17
    layers, d_model 512, 8 heads; x8 'dw_striding' sub-sampling with 256 channels, depthwise kernel 9; prediction network
    one 640-wide LSTM layer, joint 640, both with dropout 0.2; fused joint + loss in sub-batches of 4); tokenizer replaced by
    an explicit vocabulary size."""
    sizes = {"small": (176, 4, 16), "medium": (256, 4, 16), "large": (512, 8, 17)}
    d_model, n_heads, n_layers = sizes[size]
    enc = dict(feat_in=80, feat_out=-1, n_layers=n_layers, d_model=d_model, subsampling="dw_striding", subsampling_factor=8,
               subsampling_conv_channels=256, causal_downsampling=False, ff_expansion_factor=4, self_attention_model="rel_pos",
               n_heads=n_heads, att_context_size=[-1, -1], att_context_style="regular", xscaling=True, untie_biases=True,
               pos_emb_max_len=5000, conv_kernel_size=9, conv_norm_type="batch_norm", conv_context_size=None, dropout=0.1,
               dropout_pre_encoder=0.1, dropout_emb=0.0, dropout_att=0.1, stochastic_depth_drop_prob=0.0)
    enc.update(encoder_overrides)
    return {
        "sample_rate": 16000, "rnnt_reduction": "mean_batch",
        "model_defaults": dict(enc_hidden=enc["d_model"], pred_hidden=640, joint_hidden=640),
        "preprocessor": dict(sample_rate=16000, normalize="per_feature", window_size=0.025, window_stride=0.01, window="hann",
                             features=80, n_fft=512, log=True, frame_splicing=1, dither=1e-5, pad_to=0, pad_value=0.0),
        "spec_augment": dict(_target_="nemo.collections.asr.modules.SpectrogramAugmentation", freq_masks=2,
                             time_masks=10, freq_width=27, time_width=0.05) if spec_augment else None,
        "encoder": enc,
        "decoder": dict(normalization_mode=None, random_state_sampling=False, blank_as_pad=True, vocab_size=vocab_size,
                        prednet=dict(pred_hidden=640, pred_rnn_layers=1, t_max=None, dropout=0.2)),
        "joint": dict(log_softmax=None, preserve_memory=False, fuse_loss_wer=True, fused_batch_size=4, num_classes=vocab_size,
                      jointnet=dict(joint_hidden=640, activation="relu", dropout=0.2)),
        "loss": dict(loss_name="default", warprnnt_numba_kwargs=dict(fastemit_lambda=0.0, clamp=-1.0)),
        "optim": dict(name="adamw", lr=5.0, betas=[0.9, 0.98], weight_decay=1e-3,
                      sched=dict(name="NoamAnnealing", d_model=d_model, warmup_steps=10000, warmup_ratio=None, min_lr=1e-6)),
    }
