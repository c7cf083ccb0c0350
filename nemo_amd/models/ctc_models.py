"""Drop-in for `nemo.collections.asr.models.EncDecCTCModel` (models/ctc_models.py:49) restricted to the training hot
path: config-driven construction of preprocessor / encoder / decoder / loss (ctc_models.py:52-85), `forward`
(:495-546), `training_step` (:549-604, returns {'loss', 'log'}), optimizer set-up from the `optim` config section
(modelPT.py:650-823 -> fused AdamW + NoamAnnealing) and `.nemo` save / restore (modelPT.py:395,436).

Lightning is not required: `fit_step()` is the body Lightning's loop would run per batch (zero_grad -> training_step ->
backward with bucketed RCCL gradient all-reduce overlapped on a side stream -> optimizer + scheduler step)."""
from __future__ import annotations

import copy
import os
import contextlib
from typing import Any, Dict, Optional

import torch
from torch import nn

from ..core import NeuralModule, Serialization, load_nemo, resolve_target, save_nemo
from ..modules import AudioToMelSpectrogramPreprocessor, ConformerEncoder, ConvASRDecoder, CTCLoss
from ..optim import CosineAnnealing, FusedAdamW, NoamAnnealing, NoamHoldAnnealing
from ..parallel import GradSync

_DEFAULT_TARGETS = {
    "preprocessor": "nemo.collections.asr.modules.AudioToMelSpectrogramPreprocessor",
    "encoder": "nemo.collections.asr.modules.ConformerEncoder",
    "decoder": "nemo.collections.asr.modules.ConvASRDecoder",
    "spec_augment": "nemo.collections.asr.modules.SpectrogramAugmentation",
}


def _build(section: str, cfg: Dict[str, Any]):
    cfg = dict(cfg)
    cfg.setdefault("_target_", _DEFAULT_TARGETS[section])
    return Serialization.from_config_dict(cfg)


class EncDecCTCModel(nn.Module):
    def __init__(self, cfg: Dict[str, Any], trainer=None):
        super().__init__()
        cfg = copy.deepcopy(dict(cfg))
        self._cfg = cfg
        self.trainer = trainer
        self.preprocessor = _build("preprocessor", cfg["preprocessor"])
        self.encoder = _build("encoder", cfg["encoder"])
        dec = dict(cfg["decoder"])
        if dec.get("feat_in") is None:  # ctc_models.py:65-67
            dec["feat_in"] = self.encoder._feat_out
        if not dec.get("feat_in"):
            raise ValueError("param feat_in of the decoder's config is not set!")
        if dec.get("num_classes", -1) < 1 and dec.get("vocabulary") is not None:  # ctc_models.py:71-77
            dec["num_classes"] = len(dec["vocabulary"])
        self.decoder = _build("decoder", dec)
        self.loss = CTCLoss(num_classes=self.decoder.num_classes_with_blank - 1, zero_infinity=True,
                            reduction=cfg.get("ctc_reduction", "mean_batch"))
        sa = cfg.get("spec_augment")  # ctc_models.py:86-89
        self.spec_augmentation = _build("spec_augment", sa) if sa else None
        self._interctc = self._check_interctc(cfg.get("interctc"))
        if self._interctc is not None:
            bad_l = [l for l in self._interctc[1] if not (0 <= int(l) < len(self.encoder.layers))]
            if bad_l:
                raise ValueError(f"interctc.apply_at_layers {bad_l}: the encoder has {len(self.encoder.layers)} layers")
            self.encoder.capture_layers = [int(l) for l in self._interctc[1]]
        # skip_nan_grad (models/asr_model.py:147-174 on_after_backward): a step whose gradients hold NaN / Inf on ANY rank is skipped --
        # the gradients are zeroed everywhere.  The check needs the whole gradient, so the optimizer slices do not run inside backward.
        self._skip_nan_grad = bool(cfg.get("skip_nan_grad"))
        self.skipped_steps = 0
        self._optimizer: Optional[FusedAdamW] = None
        self._scheduler: Optional[NoamAnnealing] = None
        self._syncs = None
        self._wer = None
        self.validation_step_outputs, self.test_step_outputs = [], []
        # optimizer slices behind backward (begin_step / step_range / finish_step): a layer's AdamW update runs on the
        # weight-gradient stream as soon as that layer's gradients are final, instead of as one 0.9-ms launch on the main stream
        # after backward (the main stream is 98 % busy: tools/stream_gaps.py).  In-process A/B (tools/step_ab.py, round 3): 41.90 ->
        # 41.65 ms per step on one GPU; in data-parallel runs the slice sits behind its bucket's all-reduce.  With a global-norm
        # gradient clip the update needs the whole gradient first and falls back to the single launch (FusedAdamW.begin_step).
        # MI355X_OPT_IN_BACKWARD: 1 = always, 0 = never, unset = on one GPU only -- behind RCCL buckets the path has run on gloo and on
        # two ranks sharing a GPU, never on a multi-GPU node (no such node is available to this build), so data-parallel runs keep
        # the plain post-backward step until a hardware run says otherwise (round-3 advisor finding)
        # (attribute: True / False = forced, None = the rule above)
        _e = os.environ.get("MI355X_OPT_IN_BACKWARD")
        self.optimizer_in_backward = None if _e is None else (_e == "1")
        self.global_step = 0

    @staticmethod
    def _check_interctc(ic):
        """`interctc: {loss_weights: [...], apply_at_layers: [...]}` (ctc_models.py:115, parts/mixins/interctc_mixin.py:46-73).  The
        recipes ship it empty (= off).  The reference's own validation (same ValueErrors); -> (weights, layers) or None."""
        if not ic:
            return None
        weights, layers = list(ic.get("loss_weights") or []), list(ic.get("apply_at_layers") or [])
        if 1.0 - sum(weights) <= 0.0:
            raise ValueError("Make sure that sum of intermediate loss weights is < 1.0. Note that we don't do any normalization and "
                             "assign remaining weight to the regular model loss. E.g., if interctc.loss_weights = [0.1, 0.3], regular "
                             "loss will have weight of 0.6")
        if len(layers) != len(weights):
            raise ValueError("Length of interctc.apply_at_layers has to match interctc.loss_weights")
        return (weights, layers) if weights else None

    def add_interctc_losses(self, loss_value, transcript, transcript_len, encoded_len):
        """InterCTC (parts/mixins/interctc_mixin.py:214-270, ctc_models.py:577-585): the SAME decoder and CTC loss on the outputs the
        encoder captured at `apply_at_layers` (ConformerEncoder.captured, conformer_encoder.py:724-736);
        loss = (1 - sum w) * final + sum_l w_l * CTC(decoder(layer l output)).  -> (loss, metrics)"""
        if self._interctc is None:
            return loss_value, {}
        weights, layers = self._interctc
        metrics = {"final_loss": loss_value.detach()}
        loss_value = loss_value * (1.0 - sum(weights))
        for l, w in zip(layers, weights):
            cap = self.encoder.captured.get(int(l))
            if cap is None:
                raise RuntimeError(f"InterCTC: the encoder did not capture the output of layer {l}")
            inter = self.loss(log_probs=self.decoder(encoder_output=cap), targets=transcript, input_lengths=encoded_len,
                              target_lengths=transcript_len)
            metrics[f"inter_ctc_loss_l{l}"] = inter.detach()
            loss_value = loss_value + inter * w
        return loss_value, metrics

    @property
    def world_size(self) -> int:
        """queried when needed, not at construction: the process group may be initialised after the model is built (a
        constructor-time snapshot would silently train unsynchronised replicas)"""
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            return torch.distributed.get_world_size()
        return 1

    # ------------------------------------------------------------------ forward (ctc_models.py:495-546)
    def forward(self, input_signal=None, input_signal_length=None, processed_signal=None, processed_signal_length=None):
        has_input_signal = input_signal is not None and input_signal_length is not None
        has_processed_signal = processed_signal is not None and processed_signal_length is not None
        if (has_input_signal ^ has_processed_signal) is False:
            raise ValueError(f"{self} Arguments ``input_signal`` and ``input_signal_length`` are mutually exclusive "
                             " with ``processed_signal`` and ``processed_signal_len`` arguments.")
        if not has_processed_signal:
            processed_signal, processed_signal_length = self.preprocessor(input_signal=input_signal, length=input_signal_length)
        if self.spec_augmentation is not None and self.training:  # ctc_models.py:532-533
            processed_signal = self.spec_augmentation(input_spec=processed_signal, length=processed_signal_length)
        encoded, encoded_len = self.encoder(audio_signal=processed_signal, length=processed_signal_length)
        log_probs = self.decoder(encoder_output=encoded)
        greedy_predictions = log_probs.argmax(dim=-1, keepdim=False)
        return log_probs, encoded_len, greedy_predictions

    # ------------------------------------------------------------------ training_step (ctc_models.py:549-604)
    def training_step(self, batch, batch_nb=0):
        signal, signal_len, transcript, transcript_len = batch
        log_probs, encoded_len, predictions = self.forward(input_signal=signal, input_signal_length=signal_len)
        loss_value = self.loss(log_probs=log_probs, targets=transcript, input_lengths=encoded_len,
                               target_lengths=transcript_len)
        loss_value, inter_logs = self.add_interctc_losses(loss_value, transcript, transcript_len, encoded_len)
        logs = {"train_loss": loss_value.detach(), "global_step": self.global_step, **inter_logs}
        if self._scheduler is not None:
            logs["learning_rate"] = self._scheduler.get_last_lr()
        n = self._log_every_n_steps()
        if n and (batch_nb + 1) % n == 0 and self.wer is not None:  # ctc_models.py:591-600 (the decode's D2H is the host sync)
            self.wer.update(predictions=log_probs.detach(), predictions_lengths=encoded_len, targets=transcript,
                            targets_lengths=transcript_len)
            logs["training_batch_wer"] = self.wer.compute()[0]
            self.wer.reset()
        return {"loss": loss_value, "log": logs}

    def _log_every_n_steps(self):
        """`trainer.log_every_n_steps` when a trainer is attached (ctc_models.py:567-570).  Without one the reference decodes
        and scores EVERY step (a host sync per step); here that needs the explicit config key `log_every_n_steps`."""
        if self.trainer is not None and getattr(self.trainer, "log_every_n_steps", None):
            return int(self.trainer.log_every_n_steps)
        return int(self._cfg.get("log_every_n_steps") or 0)

    @property
    def wer(self):
        """greedy CTC decoding + WER over the decoder's vocabulary (ctc_models.py:91-105: CTCDecoding + WER(use_cer, ...))"""
        if self._wer is None:
            vocab = getattr(self.decoder, "vocabulary", None)
            tok = getattr(self, "tokenizer", None)
            if tok is not None and hasattr(tok, "vocab"):
                vocab = tok.vocab
            if vocab is None:
                return None
            from ..modules import WER, GreedyCTCDecoder
            strategy = str((self._cfg.get("decoding") or {}).get("strategy", "greedy_batch"))
            if strategy not in ("greedy", "greedy_batch"):  # ctc_decoding.py:231-236: beam / pyctcdecode / flashlight / wfst need their own decoders
                raise NotImplementedError(f"CTC decoding strategy '{strategy}' (implemented: greedy, greedy_batch)")
            self._wer = WER(GreedyCTCDecoder(vocabulary=list(vocab)), use_cer=bool(self._cfg.get("use_cer", False)))
        return self._wer

    # ------------------------------------------------------------------ evaluation (ctc_models.py:604-700, asr_model.py:95-160)
    @torch.no_grad()
    def validation_pass(self, batch, batch_idx=0, dataloader_idx=0):
        signal, signal_len, transcript, transcript_len = batch[:4]
        log_probs, encoded_len, _ = self.forward(input_signal=signal, input_signal_length=signal_len)
        loss_value = self.loss(log_probs=log_probs, targets=transcript, input_lengths=encoded_len, target_lengths=transcript_len)
        metrics = {"val_loss": loss_value}
        if self.wer is not None:
            self.wer.update(predictions=log_probs, predictions_lengths=encoded_len, targets=transcript,
                            targets_lengths=transcript_len)
            wer, num, denom = self.wer.compute()
            self.wer.reset()
            metrics.update({"val_wer_num": num, "val_wer_denom": denom, "val_wer": wer})
        return metrics

    def validation_step(self, batch, batch_idx=0, dataloader_idx=0):
        metrics = self.validation_pass(batch, batch_idx, dataloader_idx)
        self.validation_step_outputs.append(metrics)
        return metrics

    def test_step(self, batch, batch_idx=0, dataloader_idx=0):
        logs = self.validation_pass(batch, batch_idx, dataloader_idx)
        logs = {name.replace("val_", "test_"): value for name, value in logs.items()}
        self.test_step_outputs.append(logs)
        return logs

    def multi_validation_epoch_end(self, outputs, dataloader_idx: int = 0, prefix: str = "val"):
        """asr_model.py:95-123: mean of the batch losses, WER = sum of edit distances / sum of reference words"""
        losses = torch.stack([x[f"{prefix}_loss"] for x in outputs])
        has_wer = bool(outputs) and f"{prefix}_wer_num" in outputs[0]
        num = float(sum(x[f"{prefix}_wer_num"] for x in outputs)) if has_wer else 0.0
        denom = float(sum(x[f"{prefix}_wer_denom"] for x in outputs)) if has_wer else 0.0
        loss_sum, count = losses.sum(), float(losses.numel())
        if self.world_size > 1:
            # the DistributedSampler shards the validation set: every rank must report the metric of the WHOLE set (the
            # reference's WER torchmetric and Lightning's sync_dist loss both reduce over the ranks)
            t = torch.stack([loss_sum.double(), torch.tensor(count, device=loss_sum.device, dtype=torch.float64),
                             torch.tensor(num, device=loss_sum.device, dtype=torch.float64),
                             torch.tensor(denom, device=loss_sum.device, dtype=torch.float64)])
            torch.distributed.all_reduce(t)
            loss_sum, count, num, denom = t[0].to(losses.dtype), float(t[1]), float(t[2]), float(t[3])
        loss_mean = loss_sum / count
        logs = {f"{prefix}_loss": loss_mean}
        if has_wer:
            logs[f"{prefix}_wer"] = num / denom if denom else float("inf")
        return {f"{prefix}_loss": loss_mean, "log": logs}

    def multi_test_epoch_end(self, outputs, dataloader_idx: int = 0):
        return self.multi_validation_epoch_end(outputs, dataloader_idx, prefix="test")

    @torch.no_grad()
    def validate(self, dataloader=None):
        """one pass over the validation loader in eval mode (what the trainer's validation loop does)"""
        from ..data import DeviceBatchLoader
        dl = dataloader if dataloader is not None else getattr(self, "_validation_dl", None)
        if dl is None:
            raise RuntimeError("call setup_validation_data() first")
        was_training = self.training
        self.eval()
        self.validation_step_outputs = []
        device = next(self.parameters()).device
        try:
            for i, batch in enumerate(DeviceBatchLoader(dl, device)):
                self.validation_step(list(batch), i)
        finally:
            self.train(was_training)
        return self.multi_validation_epoch_end(self.validation_step_outputs)["log"]

    @torch.no_grad()
    def predict_step(self, batch, batch_idx=0, dataloader_idx=0):
        signal, signal_len, _, _, sample_id = batch
        log_probs, encoded_len, _ = self.forward(input_signal=signal, input_signal_length=signal_len)
        texts = self.wer.decoding(log_probs, encoded_len)
        if isinstance(sample_id, torch.Tensor):
            sample_id = sample_id.cpu().numpy()
        return list(zip(sample_id, texts))

    @torch.no_grad()
    def transcribe(self, audio, batch_size: int = 4, return_hypotheses: bool = False, num_workers: int = 0,
                   channel_selector=None, verbose: bool = False):
        """`ASRTranscriptionMixin.transcribe` (parts/mixins/transcription.py:184-290) for the greedy CTC path: `audio` is a
        path, a list of paths, or a list of 1-D waveforms (numpy / torch, at the model's sample rate); returns one string
        per input (or (text, token ids, score) triples with return_hypotheses).  Eval mode, dither and padding to 16 off
        (ctc_models.py:760-790), inputs sorted nowhere: the order of the outputs is the order of the inputs."""
        from ..data import load_audio
        if self.wer is None:
            raise RuntimeError("transcribe() needs a vocabulary (decoder.vocabulary or a tokenizer)")
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
                log_probs, enc_len, _ = self.forward(input_signal=sig.to(device), input_signal_length=lens.to(device))
                tokens, out_len, score = self.wer.decoding.decode_ids(log_probs, enc_len)
                tokens, out_len, score = tokens.cpu(), out_len.cpu(), score.cpu()
                for r in range(len(waves)):
                    ids = tokens[r, : int(out_len[r])].tolist()
                    text = self.wer.decoding.ids_to_text(ids)
                    out.append((text, ids, float(score[r])) if return_hypotheses else text)
        finally:
            feat.dither, feat.pad_to = dither, pad_to
            self.train(was_training)
        return out

    # ------------------------------------------------------------------ fine-tuning on another alphabet (ctc_models.py:190-262)
    def change_vocabulary(self, new_vocabulary, decoding_cfg=None):
        """replaces the decoder by a freshly initialised one over `new_vocabulary` (the encoder is kept), rebuilds the
        loss (blank = len(new_vocabulary)) and the decoding / WER objects, and records the labels in the dataset configs;
        optimizer state refers to the old decoder buffer, so `setup_optimization()` has to be called again"""
        if self.decoder.vocabulary == new_vocabulary:
            return
        if new_vocabulary is None or len(new_vocabulary) == 0:
            raise ValueError(f"New vocabulary must be non-empty list of chars. But I got: {new_vocabulary}")
        dec = dict(self._cfg["decoder"])
        dec.setdefault("_target_", _DEFAULT_TARGETS["decoder"])
        dec["feat_in"] = self.decoder._feat_in
        dec["vocabulary"] = list(new_vocabulary)
        dec["num_classes"] = len(new_vocabulary)
        device = next(self.decoder.parameters()).device
        compute_dtype = getattr(self.decoder, "compute_dtype", None)
        del self.decoder
        self.decoder = _build("decoder", dec).to(device)
        if compute_dtype is not None:
            self.decoder.compute_dtype = compute_dtype
        self.loss = CTCLoss(num_classes=self.decoder.num_classes_with_blank - 1, zero_infinity=True,
                            reduction=self._cfg.get("ctc_reduction", "mean_batch"))
        self._wer = None            # rebuilt lazily over the new vocabulary
        self._optimizer = self._scheduler = self._syncs = None
        self._cfg["decoder"] = dec
        self._cfg["labels"] = list(new_vocabulary)
        for key in ("train_ds", "validation_ds", "test_ds"):
            if isinstance(self._cfg.get(key), dict):
                self._cfg[key]["labels"] = list(new_vocabulary)

    def setup_test_data(self, test_data_config: Dict[str, Any]):
        test_data_config = dict(test_data_config)
        test_data_config.setdefault("shuffle", False)
        self._cfg["test_ds"] = {k: v for k, v in test_data_config.items() if k != "tokenizer" or isinstance(v, dict)}
        self._test_dl = self._setup_dataloader_from_config(test_data_config)
        return self._test_dl

    def test_dataloader(self):
        return getattr(self, "_test_dl", None)

    # ------------------------------------------------------------------ optimisation
    def trainable_modules(self):
        """the modules that own flat parameter buffers, in backward-completion order for the gradient exchange"""
        return [self.encoder, self.decoder]

    def flats(self):
        return [m.flat_parameters() for m in self.trainable_modules()]

    def setup_optimization(self, optim_config: Optional[Dict[str, Any]] = None):
        oc = dict(optim_config if optim_config is not None else self._cfg.get("optim", {}))
        name = oc.get("name", "adamw")
        if name != "adamw":
            raise NotImplementedError(f"optimizer '{name}': the Conformer-CTC recipes use adamw")
        # `gradient_clip_val` is the trainer's key (conformer_ctc_bpe.yaml:204), `ema.decay` the EMA callback's
        # (exp_manager.ema, nemo/collections/common/callbacks/ema.py:27-62); both ride inside the fused AdamW launch
        ema = oc.get("ema") or {}
        self._optimizer = FusedAdamW(self.flats(), lr=oc.get("lr", 1e-3), betas=tuple(oc.get("betas", (0.9, 0.999))),
                                     eps=oc.get("eps", 1e-8), weight_decay=oc.get("weight_decay", 0.0),
                                     max_grad_norm=oc.get("gradient_clip_val") or None,
                                     ema_decay=ema.get("decay") if ema.get("enable", bool(ema)) else None)
        sched = oc.get("sched")
        if sched:
            if sched.get("name") == "NoamHoldAnnealing":  # the Squeezeformer recipe (squeezeformer_ctc_bpe.yaml:160-168)
                self._scheduler = NoamHoldAnnealing(oc.get("lr", 1e-3), warmup_steps=sched.get("warmup_steps"),
                                                    warmup_ratio=sched.get("warmup_ratio"), hold_steps=sched.get("hold_steps"),
                                                    hold_ratio=sched.get("hold_ratio"), max_steps=sched.get("max_steps"),
                                                    decay_rate=sched.get("decay_rate", 0.5), min_lr=sched.get("min_lr", 0.0))
                return self._optimizer, self._scheduler
            if sched.get("name") == "CosineAnnealing":  # the FastConformer recipes (fast-conformer_*_bpe.yaml)
                max_steps = sched.get("max_steps") or (getattr(self.trainer, "max_steps", None) if self.trainer is not None else None)
                if not max_steps or max_steps < 0:
                    # modelPT.py prepare_lr_scheduler: without `max_steps` (or a dataloader to derive it from) the reference
                    # logs a warning and trains WITHOUT a scheduler
                    import warnings
                    warnings.warn("CosineAnnealing needs `max_steps` (optim.sched.max_steps or trainer.max_steps): "
                                  "scheduler will not be instantiated")
                    self._scheduler = None
                    return self._optimizer, self._scheduler
                self._scheduler = CosineAnnealing(oc.get("lr", 1e-3), max_steps=max_steps, warmup_steps=sched.get("warmup_steps"),
                                                  warmup_ratio=sched.get("warmup_ratio"), constant_steps=sched.get("constant_steps"),
                                                  constant_ratio=sched.get("constant_ratio"), min_lr=sched.get("min_lr", 0.0))
                return self._optimizer, self._scheduler
            if sched.get("name") != "NoamAnnealing":
                raise NotImplementedError(f"scheduler '{sched.get('name')}'")
            self._scheduler = NoamAnnealing(oc.get("lr", 1e-3), d_model=sched["d_model"], warmup_steps=sched.get("warmup_steps"),
                                            warmup_ratio=sched.get("warmup_ratio"), max_steps=sched.get("max_steps"),
                                            min_lr=sched.get("min_lr", 0.0))
        return self._optimizer, self._scheduler

    def _grad_syncs(self):
        gens = tuple(m.flat_parameters().generation for m in self.trainable_modules())
        if self._syncs is not None and getattr(self, "_syncs_gen", None) != gens:
            self._syncs = None  # a flat buffer was rebuilt (model.to(), ...): the exchange must not reduce the stale one
        if self._syncs is None:
            self._syncs_gen = gens
            self._syncs = []
            for mod in self.trainable_modules():
                gs = GradSync(mod.flat_parameters().grad)
                mod.grad_ready_hook = gs.ready
                if gs.use_side_stream and hasattr(mod, "_wg_stream"):  # weight gradients are produced on their own stream
                    gs.producer_streams = (lambda m=mod: [m._wg_stream] if m._wg_stream is not None else [])
                    mod._wgrad_join_per_layer = False
                if hasattr(mod, "setup_process_groups"):
                    mod.setup_process_groups()  # (collective when the own-group option is on: every rank is here, before step 1)
                self._syncs.append(gs)
        return self._syncs

    def fit_step(self, batch):
        """one optimizer step = what Lightning's loop does per batch with trainer.strategy=ddp"""
        if self._optimizer is None:
            self.setup_optimization()
        syncs = self._grad_syncs() if self.world_size > 1 else []
        self._optimizer.zero_grad()
        # Lightning order (optimizer.step(), then scheduler.step() with interval 'step'): optimizer step n runs with
        # lr(max(1, n - 1)) of the Noam formula (lr_scheduler.py:518-576 reads `last_epoch` before it is advanced)
        lr = self._scheduler.get_last_lr() if self._scheduler is not None else None
        scale = syncs[0].grad_scale if syncs else 1.0 / self.world_size  # (1 when the buckets travel pre-scaled as bf16)
        use_early = (not syncs) if self.optimizer_in_backward is None else bool(self.optimizer_in_backward)
        if self._skip_nan_grad:
            use_early = False
        early = use_early and self._optimizer.begin_step(lr=lr, grad_scale=scale)
        if early:  # slices of the flat buffers are updated as soon as their gradients are final (and reduced)
            self._install_early_step(syncs)
        scope = getattr(self.encoder, "step_scope", None)
        with (scope() if scope is not None else contextlib.nullcontext()):  # forward -> backward -> next forward discipline
            out = self.training_step(batch, self.global_step)
            out["loss"].backward()
        self._after_backward()
        for gs in syncs:
            gs.wait()
        skip = self._skip_nan_grad and not self.on_after_backward()
        if skip:
            pass  # (the reference's zero_grad() leaves every .grad None: its optimizer.step() then touches no parameter and no moment)
        elif early:
            if not syncs:
                self.encoder._wgrad_join()  # the last slices were updated on the weight-gradient stream
            self._optimizer.finish_step()
        else:
            if use_early:
                self._optimizer.step_count -= 1  # begin_step counted it; step() counts again
            self._optimizer.step(lr=lr, grad_scale=scale)
        if self._scheduler is not None:
            self._scheduler.step()
        self.global_step += 1
        mb = getattr(self.encoder, "_syncbn_mailbox", None)
        if mb is not None:
            mb.poll()  # non-blocking: raises once a SyncBatchNorm exchange of an earlier step is known to have timed out
        return out

    def _after_backward(self):
        """streams other than the current one that produced gradients are joined here (none in the CTC model)"""

    def on_after_backward(self):
        """models/asr_model.py:147-174: zero the gradients of a step if any of them (on any rank) holds NaN / Inf.  -> True when the
        gradients are valid; False = the step is skipped (fit_step does not run the optimizer: the reference's zero_grad() sets every
        .grad to None, which torch's optimizers skip -- parameters and moments stay as they are)"""
        enc = self.encoder
        if hasattr(enc, "_wgrad_join"):
            enc._wgrad_join()  # the weight-gradient stream has written its last gradients
        flats = self.flats()
        ok = torch.ones(1, device=flats[0].grad.device, dtype=torch.float32)
        for fp in flats:
            ok = ok * torch.isfinite(fp.grad).all().to(torch.float32)
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
        if float(ok.item()) < 1.0:
            import logging
            logging.getLogger(__name__).warning("detected inf or nan values in gradients! Setting gradients to zero.")
            for fp in flats:
                fp.zero_grad()
            self.skipped_steps += 1
            return False
        return True

    def _install_early_step(self, syncs):
        opt = self._optimizer
        if syncs:
            for gs, mod in zip(syncs, self.trainable_modules()):
                fp = mod.flat_parameters()
                gs.after_reduce = (lambda lo, hi, fp=fp: opt.step_range(fp, lo, hi))
        else:
            enc = self.encoder
            fp = enc.flat_parameters()

            def hook(lo, hi, fp=fp, enc=enc):
                with enc._wgrad_scope(fp.grad):  # on the weight-gradient stream, behind the layer's grouped wgrad
                    opt.step_range(fp, lo, hi)
            enc.grad_ready_hook = hook
            enc._wgrad_join_per_layer = False  # the hook orders itself behind both streams

    # ------------------------------------------------------------------ data (ctc_models.py:303-380, 382-470)
    def _dataset_from_config(self, config: Dict[str, Any]):
        """audio_to_text_dataset.get_char_dataset / get_bpe_dataset (audio_to_text_dataset.py:132-163, 215-243): a
        `tokenizer` entry (an object with `text_to_ids`, or {'model_path': <sentencepiece .model>}) selects the BPE dataset
        (EncDecCTCModelBPE), `labels` the character dataset"""
        from ..data import AudioToBPEDataset, AudioToCharDataset, SentencePieceTokenizer
        config.setdefault("sample_rate", self._cfg.get("sample_rate", 16000))  # inject_dataloader_value_from_model_config
        if config.get("labels") is None and self._cfg.get("labels") is not None:
            config["labels"] = self._cfg["labels"]
        tok = config.get("tokenizer", getattr(self, "tokenizer", None))
        if tok is not None:
            if isinstance(tok, dict):
                tok = SentencePieceTokenizer(tok["model_path"])
            self.tokenizer = tok
            return AudioToBPEDataset(
                manifest_filepath=config["manifest_filepath"], tokenizer=tok, sample_rate=config["sample_rate"],
                int_values=config.get("int_values", False), max_duration=config.get("max_duration"),
                min_duration=config.get("min_duration"), max_utts=config.get("max_utts", 0),
                trim=config.get("trim_silence", False), use_start_end_token=config.get("use_start_end_token", True),
                return_sample_id=config.get("return_sample_id", False), channel_selector=config.get("channel_selector"))
        return AudioToCharDataset(
            manifest_filepath=config["manifest_filepath"], labels=config.get("labels"), sample_rate=config["sample_rate"],
            int_values=config.get("int_values", False), max_duration=config.get("max_duration"),
            min_duration=config.get("min_duration"), max_utts=config.get("max_utts", 0),
            blank_index=config.get("blank_index", -1), unk_index=config.get("unk_index", -1),
            normalize=config.get("normalize_transcripts", False), trim=config.get("trim_silence", False),
            parser=config.get("parser", "base"), return_sample_id=config.get("return_sample_id", False),
            channel_selector=config.get("channel_selector"))

    def _setup_dataloader_from_config(self, config: Dict[str, Any]):
        from ..data import SemiSortBatchSampler
        config = dict(config)
        # data-set kinds and augmentations this input side does not provide must not be skipped silently: they change WHAT is trained on
        # (audio_to_text_dataset.py:215-330: tarred / concat / lhotse datasets; perturb.py `augmentor`: speed, noise, impulse, ...)
        unsupported = [k for k in ("is_tarred", "use_lhotse", "is_concat") if config.get(k)] + \
                      (["augmentor"] if config.get("augmentor") else [])
        if unsupported:
            raise NotImplementedError("data-set options not provided by the MI355X input pipeline: " + ", ".join(unsupported) +
                                      " (manifest + wav / npy data sets: AudioToCharDataset / AudioToBPEDataset)")
        if config.get("manifest_filepath") is None:
            return None
        dataset = self._dataset_from_config(config)
        rank = torch.distributed.get_rank() if self.world_size > 1 else 0
        shuffle, sampler, batch_size = config["shuffle"], None, config["batch_size"]
        drop_last = config.get("drop_last", False)
        if config.get("use_semi_sorted_batching", False):
            # batches are shaped by duration and dealt to the ranks by the sampler; automatic batching is off
            # (ctc_models.py:355-366, asr_batching.py:204-240)
            sampler = SemiSortBatchSampler(global_rank=rank, world_size=self.world_size, durations=dataset.durations,
                                           batch_size=batch_size, batch_shuffle=config.get("shuffle", True),
                                           drop_last=drop_last, randomization_factor=config.get("randomization_factor"),
                                           seed=config.get("semi_sort_sampler_seed", 42),
                                           synced_rng=config.get("semi_sort_synced_rng", False))
            batch_size, drop_last, shuffle = None, False, False
        elif self.world_size > 1:  # what Lightning injects under trainer.strategy=ddp
            sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=self.world_size, rank=rank,
                                                                      shuffle=shuffle, drop_last=drop_last)
            shuffle = False
        return torch.utils.data.DataLoader(dataset=dataset, batch_size=batch_size, sampler=sampler, batch_sampler=None,
                                           collate_fn=dataset._collate_fn, drop_last=drop_last, shuffle=shuffle,
                                           num_workers=config.get("num_workers", 0), pin_memory=False)

    def setup_training_data(self, train_data_config: Dict[str, Any]):
        train_data_config = dict(train_data_config)
        train_data_config.setdefault("shuffle", True)
        self._cfg["train_ds"] = {k: v for k, v in train_data_config.items() if k != "tokenizer" or isinstance(v, dict)}
        self._train_dl = self._setup_dataloader_from_config(train_data_config)
        return self._train_dl

    def setup_validation_data(self, val_data_config: Dict[str, Any]):
        val_data_config = dict(val_data_config)
        val_data_config.setdefault("shuffle", False)
        self._cfg["validation_ds"] = {k: v for k, v in val_data_config.items() if k != "tokenizer" or isinstance(v, dict)}
        self._validation_dl = self._setup_dataloader_from_config(val_data_config)
        return self._validation_dl

    def train_dataloader(self):
        return getattr(self, "_train_dl", None)

    def fit(self, max_steps: int, max_epochs: int = 1 << 30, log_every: int = 0):
        """the trainer loop for this path: batches staged through pinned memory and copied on their own HIP stream while
        the previous step computes (`data.DeviceBatchLoader`), then `fit_step`.  Returns the per-step losses (device
        scalars: nothing here synchronises the host with the GPU unless `log_every` asks for a printout)."""
        from ..data import DeviceBatchLoader
        dl = self.train_dataloader()
        if dl is None:
            raise RuntimeError("call setup_training_data() first")
        device = next(self.parameters()).device
        losses = []
        for epoch in range(max_epochs):
            if hasattr(dl.sampler, "set_epoch"):
                dl.sampler.set_epoch(epoch)
            for batch in DeviceBatchLoader(dl, device):
                losses.append(self.fit_step(list(batch[:4]))["loss"].detach())
                if log_every and len(losses) % log_every == 0:
                    print(f"step {len(losses)} loss {losses[-1].item():.4f}", flush=True)
                if len(losses) >= max_steps:
                    return losses
        return losses

    # ------------------------------------------------------------------ .nemo (modelPT.py:395,436)
    def _artifacts(self) -> Dict[str, str]:
        return {}

    def save_to(self, save_path: str):
        save_nemo(save_path, dict(self._cfg, target=f"{type(self).__module__}.{type(self).__name__}"), self.state_dict(),
                  artifacts=self._artifacts())

    @classmethod
    def restore_from(cls, restore_path: str, map_location=None, strict: bool = True):
        cfg, sd = load_nemo(restore_path)
        cfg.pop("target", None)
        model = cls(cfg)
        if map_location is not None:
            model = model.to(map_location)
        model.load_state_dict(sd, strict=strict)
        model.encoder.weights_updated(); model.decoder.weights_updated()
        return model


class EncDecCTCModelBPE(EncDecCTCModel):
    """Drop-in for `nemo.collections.asr.models.EncDecCTCModelBPE` (ctc_bpe_models.py:39-110), the class the Conformer-CTC
    BPE recipe instantiates (`examples/asr/asr_ctc/speech_to_text_ctc_bpe.py`): the `tokenizer` section ({dir, type: bpe} or
    {model_path}) selects a SentencePiece model (`ASRBPEMixin._setup_tokenizer`, parts/mixins/mixins.py:60-190), whose
    pieces in id order become the decoder vocabulary; a placeholder `num_classes` (< 1) is replaced by their number.
    The tokenizer model travels inside the `.nemo` file as an artifact."""

    def __init__(self, cfg: Dict[str, Any], trainer=None):
        from ..data import SentencePieceTokenizer
        cfg = copy.deepcopy(dict(cfg))
        if "tokenizer" not in cfg:
            raise ValueError("`cfg` must have `tokenizer` config to create a tokenizer !")
        tok_cfg = dict(cfg["tokenizer"])
        tok_type = str(tok_cfg.get("type", "bpe")).lower()
        if tok_type not in ("bpe", "wpe"):
            raise ValueError("`tokenizer.type` must be either `bpe` for SentencePiece tokenizer or `wpe` for BERT based "
                             "tokenizer")
        if tok_type == "wpe":
            raise NotImplementedError("WordPiece (BERT) tokenizers: the Conformer-CTC recipes use SentencePiece `bpe`")
        if tok_cfg.get("special_tokens") is not None:
            raise ValueError("`special_tokens` are no longer supported for SentencePiece based tokenizers.")
        model_path = tok_cfg.get("model_path") or os.path.join(tok_cfg["dir"], "tokenizer.model")
        tokenizer = SentencePieceTokenizer(model_path)
        vocabulary = tokenizer.vocab
        dec = dict(cfg["decoder"])
        dec["vocabulary"] = vocabulary
        if dec.get("num_classes", -1) < 1:
            dec["num_classes"] = len(vocabulary)
        cfg["decoder"] = dec
        cfg["tokenizer"] = dict(tok_cfg, model_path=model_path, type="bpe")
        super().__init__(cfg, trainer=trainer)
        self.tokenizer = tokenizer
        self.tokenizer_type = "bpe"

    def _artifacts(self):
        return {"tokenizer.model_path": self._cfg["tokenizer"]["model_path"]}

    def change_vocabulary(self, new_tokenizer_dir, new_tokenizer_type: str = "bpe", decoding_cfg=None):
        """ctc_bpe_models.py:113-210: a new tokenizer directory instead of a list of labels"""
        from ..data import SentencePieceTokenizer
        if new_tokenizer_type.lower() != "bpe":
            raise ValueError("New tokenizer type must be either `bpe` or `wpe`" if new_tokenizer_type.lower() != "wpe"
                             else "WordPiece tokenizers are not provided")
        model_path = new_tokenizer_dir if os.path.isfile(new_tokenizer_dir) else os.path.join(new_tokenizer_dir, "tokenizer.model")
        if not os.path.isfile(model_path):
            raise NotADirectoryError(f"New tokenizer dir must be non-empty path to a directory. But I got: {new_tokenizer_dir}")
        tokenizer = SentencePieceTokenizer(model_path)
        super().change_vocabulary(tokenizer.vocab)
        self.tokenizer = tokenizer
        self._wer = None
        self._cfg["tokenizer"] = dict(self._cfg.get("tokenizer", {}), dir=os.path.dirname(model_path), model_path=model_path,
                                      type="bpe")


def squeezeformer_ctc_config(size: str = "medium", vocab_size: int = 128, spec_augment: bool = False,
                             **encoder_overrides) -> Dict[str, Any]:
    """model section of examples/asr/conf/squeezeformer/squeezeformer_ctc_bpe.yaml (:22-168) for the sizes of its table (:9-16:
    d_model, n_layers, n_heads, SpecAugment time masks, peak lr, time_reduce_idx); BASELINE.json configs[4] is 'medium'."""
    sizes = {"xs": (144, 16, 4, 5, 2e-3, 7), "small": (196, 18, 4, 5, 2e-3, 8), "sm": (256, 16, 4, 5, 1.5e-3, 7),
             "medium": (324, 20, 4, 7, 1.5e-3, 9), "ml": (512, 18, 8, 10, 1e-3, 8), "large": (640, 22, 8, 10, 5e-4, 10)}
    d_model, n_layers, n_heads, time_masks, lr, reduce_idx = sizes[size]
    enc = dict(_target_="nemo.collections.asr.modules.SqueezeformerEncoder", feat_in=80, feat_out=-1, n_layers=n_layers,
               d_model=d_model, adaptive_scale=True, time_reduce_idx=reduce_idx, time_recovery_idx=None, subsampling="dw_striding",
               subsampling_factor=4, subsampling_conv_channels=-1, ff_expansion_factor=4, self_attention_model="rel_pos",
               n_heads=n_heads, att_context_size=[-1, -1], xscaling=True, untie_biases=True, pos_emb_max_len=5000,
               conv_kernel_size=31, conv_norm_type="batch_norm", dropout=0.1, dropout_emb=0.0, dropout_att=0.1)
    enc.update(encoder_overrides)
    return {
        "sample_rate": 16000, "ctc_reduction": "mean_batch", "skip_nan_grad": False,
        "preprocessor": dict(sample_rate=16000, normalize="per_feature", window_size=0.025, window_stride=0.01, window="hann",
                             features=80, n_fft=512, log=True, frame_splicing=1, dither=1e-5, pad_to=0, pad_value=0.0),
        "spec_augment": dict(_target_="nemo.collections.asr.modules.SpectrogramAugmentation", freq_masks=2,
                             time_masks=time_masks, freq_width=27, time_width=0.05) if spec_augment else None,
        "encoder": enc,
        "decoder": dict(feat_in=None, num_classes=vocab_size, vocabulary=None),
        "optim": dict(name="adamw", lr=lr, betas=[0.9, 0.98], weight_decay=4e-5,
                      sched=dict(name="NoamHoldAnnealing", warmup_steps=5000, warmup_ratio=None, hold_steps=40000,
                                 hold_ratio=None, decay_rate=1.0, min_lr=1e-5)),
    }


def conformer_ctc_config(size: str = "large", vocab_size: int = 128, spec_augment: bool = False,
                         **encoder_overrides) -> Dict[str, Any]:
    """model section of examples/asr/conf/conformer/conformer_ctc_bpe.yaml (:44-192) for the sizes of its table (:7-17);
    tokenizer replaced by an explicit vocabulary size.  `spec_augment=True` adds the recipe's section (:108-114:
    2 frequency masks of width <= 27, 10 time masks of width <= 5 % of the utterance)."""
    sizes = {"small": (176, 4, 16), "medium": (256, 4, 18), "large": (512, 8, 18)}
    d_model, n_heads, n_layers = sizes[size]
    enc = dict(feat_in=80, feat_out=-1, n_layers=n_layers, d_model=d_model, subsampling="striding", subsampling_factor=4,
               subsampling_conv_channels=-1, causal_downsampling=False, ff_expansion_factor=4, self_attention_model="rel_pos",
               n_heads=n_heads, att_context_size=[-1, -1], att_context_style="regular", xscaling=True, untie_biases=True,
               pos_emb_max_len=5000, conv_kernel_size=31, conv_norm_type="batch_norm", conv_context_size=None, dropout=0.1,
               dropout_pre_encoder=0.1, dropout_emb=0.0, dropout_att=0.1, stochastic_depth_drop_prob=0.0,
               stochastic_depth_mode="linear", stochastic_depth_start_layer=1)
    enc.update(encoder_overrides)
    return {
        "sample_rate": 16000, "ctc_reduction": "mean_batch", "skip_nan_grad": False,
        "preprocessor": dict(sample_rate=16000, normalize="per_feature", window_size=0.025, window_stride=0.01, window="hann",
                             features=80, n_fft=512, log=True, frame_splicing=1, dither=1e-5, pad_to=0, pad_value=0.0),
        "spec_augment": dict(_target_="nemo.collections.asr.modules.SpectrogramAugmentation", freq_masks=2,
                             time_masks=10, freq_width=27, time_width=0.05) if spec_augment else None,
        "encoder": enc,
        "decoder": dict(feat_in=None, num_classes=vocab_size, vocabulary=None),
        "optim": dict(name="adamw", lr=2.0, betas=[0.9, 0.98], weight_decay=1e-3,
                      sched=dict(name="NoamAnnealing", d_model=d_model, warmup_steps=10000, warmup_ratio=None, min_lr=1e-6)),
    }
