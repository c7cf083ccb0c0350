"""TEST INFRASTRUCTURE ONLY -- import shim that loads the reference's own source files.

Loads `FilterbankFeatures` and `ConformerEncoder` *verbatim* from /root/reference (read-only)
without running the heavy package `__init__`s that need hydra / lightning / librosa / wrapt
(none are installed; no network).  Recipe = SURVEY.md Appendix B.

Only usable inside the build container (where /root/reference exists).  It is used by
`oracle/make_golden.py` to produce the committed fixtures under tests/golden/ and by
CPU tests that pin `oracle/conformer_ref.py` (the travelling restatement) to the real code.
Nothing in the product (`nemo_amd/`) may import this file.
"""
from __future__ import annotations

import contextlib
import dataclasses
import importlib
import math
import os
import sys
import types

import numpy as np
import torch
from torch import nn

REF_ROOT = os.environ.get("NEMO_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "nemo", "collections", "asr"))


# ---------------------------------------------------------------------------------------------
# librosa.filters.mel restatement (third-party, NOT under /root/reference: librosa>=0.10.1,
# requirements/requirements_asr.txt).  Published algorithm: Slaney mel scale (htk=False),
# triangular filters on rfftfreq, area normalisation 2/(f[i+2]-f[i]).  Call site:
# nemo/collections/asr/parts/preprocessing/features.py:338-344.  "parity unpinned": the
# reference tree holds no golden filterbank values.
# ---------------------------------------------------------------------------------------------
def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    with np.errstate(divide="ignore", invalid="ignore"):
        log_t = f >= min_log_hz
        mels = np.where(log_t, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)
    return mels


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    log_t = m >= min_log_mel
    freqs = np.where(log_t, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)
    return freqs


def slaney_mel_filterbank(sr, n_fft, n_mels, fmin=0.0, fmax=None, norm="slaney"):
    if fmax is None:
        fmax = sr / 2.0
    n_bins = 1 + n_fft // 2
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_pts = np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2)
    mel_f = _mel_to_hz(mel_pts)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, n_bins), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    if norm == "slaney":
        enorm = 2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels])
        weights *= enorm[:, np.newaxis]
    return weights.astype(np.float32)


_INSTALLED = False


def _stub(name, path=None, **attrs):
    m = types.ModuleType(name)
    if path is not None:
        m.__path__ = [path]
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


def install():
    """Register the fakes + namespace stubs; idempotent."""
    global _INSTALLED
    if _INSTALLED:
        return
    if not reference_available():
        raise RuntimeError(f"reference tree not found under {REF_ROOT}")
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import nemo  # noqa: F401  (imports cleanly)
    import nemo.utils  # noqa: F401

    # ---- fake third-party modules -------------------------------------------------------
    if "omegaconf" not in sys.modules:

        class DictConfig(dict):
            pass

        class ListConfig(list):
            pass

        @contextlib.contextmanager
        def open_dict(cfg):
            yield cfg

        class OmegaConf:
            @staticmethod
            def create(x=None):
                return DictConfig(x or {})

            @staticmethod
            def to_container(x, resolve=True):
                return x

            @staticmethod
            def structured(obj):   # (a dataclass instance stays itself: attribute access is all the callers use)
                return obj

        _stub("omegaconf", DictConfig=DictConfig, ListConfig=ListConfig, open_dict=open_dict,
              OmegaConf=OmegaConf, MISSING="???")
    if "librosa" not in sys.modules:
        lib = _stub("librosa")
        _stub("librosa.filters", mel=slaney_mel_filterbank)
        lib.filters = sys.modules["librosa.filters"]

    base = os.path.join(REF_ROOT, "nemo")
    # ---- namespace stubs (heavy __init__ never runs) ------------------------------------
    for name, rel in [
        ("nemo.core", "core"),
        ("nemo.collections", "collections"),
        ("nemo.collections.asr", "collections/asr"),
        ("nemo.collections.asr.parts", "collections/asr/parts"),
        ("nemo.collections.asr.modules", "collections/asr/modules"),
        ("nemo.collections.asr.models", "collections/asr/models"),
        ("nemo.collections.asr.losses", "collections/asr/losses"),
        ("nemo.collections.asr.parts.submodules", "collections/asr/parts/submodules"),
        ("nemo.collections.asr.parts.submodules.adapters", "collections/asr/parts/submodules/adapters"),
        ("nemo.collections.asr.parts.utils", "collections/asr/parts/utils"),
        ("nemo.collections.asr.parts.mixins", "collections/asr/parts/mixins"),
        ("nemo.collections.asr.parts.preprocessing", "collections/asr/parts/preprocessing"),
        ("nemo.collections.common", "collections/common"),
        ("nemo.collections.common.parts", "collections/common/parts"),
    ]:
        _stub(name, os.path.join(base, rel))

    importlib.import_module("nemo.core.neural_types")  # real, pure python

    # ---- fake nemo.core.classes ----------------------------------------------------------
    class typecheck:  # no-op decorator with the attributes call-sites touch
        def __init__(self, *a, **k):
            pass

        def __call__(self, fn):
            return fn

        @staticmethod
        @contextlib.contextmanager
        def disable_checks():
            yield

        @staticmethod
        def set_typecheck_enabled(enabled=True):
            pass

    class Typing:
        pass

    class Serialization:
        pass

    class FileIO:
        pass

    class Exportable:
        pass

    class NeuralModule(nn.Module, Typing, Serialization, FileIO):
        pass

    class AccessMixin:
        @classmethod
        def is_access_enabled(cls, guid=None):
            return False

        def register_accessible_tensor(self, name, tensor):
            pass

    class AdapterModuleMixin:
        def is_adapter_available(self):
            return False

        def set_accepted_adapter_types(self, x):
            pass

        def forward_enabled_adapters(self, x):
            return x

    class AdapterModelPTMixin:
        pass

    def get_registered_adapter(cls):
        return None

    def register_adapter(base_class, adapter_class):
        pass

    _stub("nemo.core.classes", os.path.join(base, "core/classes"), typecheck=typecheck, Typing=Typing,
          Serialization=Serialization, FileIO=FileIO, Exportable=Exportable, NeuralModule=NeuralModule)
    _stub("nemo.core.classes.common", typecheck=typecheck, Typing=Typing, Serialization=Serialization, FileIO=FileIO)
    _stub("nemo.core.classes.exportable", Exportable=Exportable)
    _stub("nemo.core.classes.module", NeuralModule=NeuralModule)
    am = _stub("nemo.core.classes.mixins.adapter_mixins", AdapterModuleMixin=AdapterModuleMixin,
               AdapterModelPTMixin=AdapterModelPTMixin, get_registered_adapter=get_registered_adapter,
               register_adapter=register_adapter)
    _stub("nemo.core.classes.mixins", AccessMixin=AccessMixin, adapter_mixins=am,
          AdapterModuleMixin=AdapterModuleMixin)
    sys.modules["nemo.core.classes.mixins.adapter_mixins"] = am
    sys.modules["nemo.core.classes.mixins"].adapter_mixins = am
    # what modules/rnnt.py and rnnt_abstract.py import from the package roots
    sys.modules["nemo.core.classes"].adapter_mixins = am
    sys.modules["nemo.core"].NeuralModule = NeuralModule

    class Loss(nn.Module, Typing):  # nemo/core/classes/loss.py: an nn.Module with typed I/O
        pass

    sys.modules["nemo.core.classes"].Loss = Loss

    # ---- misc fakes ------------------------------------------------------------------------
    @dataclasses.dataclass
    class CacheAwareStreamingConfig:  # field list = asr_models_config.py:119-141
        chunk_size: int = 0
        shift_size: int = 0
        cache_drop_size: int = 0
        last_channel_cache_size: int = 0
        valid_out_len: int = 0
        pre_encode_cache_size: int = 0
        drop_extra_pre_encoded: int = 0
        last_channel_num: int = 0
        last_time_num: int = 0

    _stub("nemo.collections.asr.models.configs", CacheAwareStreamingConfig=CacheAwareStreamingConfig)
    _stub("nemo.collections.asr.parts.utils.adapter_utils",
          LINEAR_ADAPTER_CLASSPATH="", MHA_ADAPTER_CLASSPATH="", RELMHA_ADAPTER_CLASSPATH="",
          POS_ENCODING_ADAPTER_CLASSPATH="", REL_POS_ENCODING_ADAPTER_CLASSPATH="",
          update_adapter_cfg_input_dim=lambda *a, **k: None)

    class AudioAugmentor:
        pass

    class AudioSegment:
        pass

    _stub("nemo.collections.asr.parts.preprocessing.perturb", AudioAugmentor=AudioAugmentor)
    _stub("nemo.collections.asr.parts.preprocessing.segment", AudioSegment=AudioSegment)

    # ---- what parts/submodules/rnnt_greedy_decoding.py imports besides the transducer modules: language-model fusion and the
    # label-looping computers (both pull Lightning / CUDA-graph helpers).  Placeholders: the fixtures run the frame-looping
    # algorithm (`loop_labels=False`, :804-990), which the reference documents as producing the same hypotheses.
    class _Unavailable:
        def __init__(self, *a, **k):
            raise RuntimeError("not available through the oracle shim")

    _stub("nemo.collections.asr.parts.context_biasing", BoostingTreeModelConfig=_Unavailable, GPUBoostingTreeModel=_Unavailable)
    _stub("nemo.collections.asr.parts.submodules.ngram_lm", NGramGPULanguageModel=_Unavailable)
    _stub("nemo.collections.asr.parts.submodules.transducer_decoding", GreedyBatchedRNNTLabelLoopingComputer=_Unavailable,
          GreedyBatchedTDTLabelLoopingComputer=_Unavailable)
    _INSTALLED = True


def load_reference():
    """Returns (FilterbankFeatures, ConformerEncoder) -- the reference's own classes."""
    install()
    feats = importlib.import_module("nemo.collections.asr.parts.preprocessing.features")
    enc = importlib.import_module("nemo.collections.asr.modules.conformer_encoder")
    return feats.FilterbankFeatures, enc.ConformerEncoder


class ReferenceCTCModel(nn.Module):
    """Reference forward path (ctc_models.py:495-546) assembled from the reference's own
    FilterbankFeatures + ConformerEncoder, with the 3-line restatements of ConvASRDecoder
    (conv_asr.py:445-468) and CTCLoss (losses/ctc.py:45-82) named in SURVEY.md section 8(c)."""

    def __init__(self, d_model, n_heads, n_layers, vocab=128, feat_in=80, dropout=0.0, dropout_att=0.0,
                 dither=0.0, conv_kernel_size=31, **encoder_kwargs):
        super().__init__()
        FilterbankFeatures, ConformerEncoder = load_reference()
        self.featurizer = FilterbankFeatures(
            sample_rate=16000, n_window_size=400, n_window_stride=160, nfilt=feat_in, n_fft=512,
            dither=dither, pad_to=0, normalize="per_feature")
        self.encoder = ConformerEncoder(
            feat_in=feat_in, n_layers=n_layers, d_model=d_model, n_heads=n_heads, subsampling="striding",
            subsampling_factor=4, conv_kernel_size=conv_kernel_size, dropout=dropout,
            dropout_pre_encoder=dropout, dropout_emb=0.0, dropout_att=dropout_att, **encoder_kwargs)
        self.decoder_layers = nn.Sequential(nn.Conv1d(d_model, vocab + 1, kernel_size=1, bias=True))
        nn.init.xavier_uniform_(self.decoder_layers[0].weight)
        self.vocab = vocab
        self.ctc = nn.CTCLoss(blank=vocab, reduction="none", zero_infinity=True)

    def features(self, audio, audio_len):
        return self.featurizer(audio, audio_len)

    def forward(self, audio, audio_len, tokens, token_len):
        mel, mel_len = self.featurizer(audio.clone(), audio_len)
        enc, enc_len = self.encoder.forward(audio_signal=mel, length=mel_len)
        logp = torch.log_softmax(self.decoder_layers(enc).transpose(1, 2), dim=-1)
        loss = self.ctc(logp.transpose(1, 0), tokens.long(), enc_len.long(), token_len.long()).mean()
        return loss, logp, enc, enc_len, mel, mel_len

    def forward_interctc(self, audio, audio_len, tokens, token_len, apply_at_layers, loss_weights):
        """The reference encoder's OWN capture code (conformer_encoder.py:724-736) switched on through the three AccessMixin
        members it touches, then the loss assembly of InterCTCMixin.add_interctc_losses (parts/mixins/interctc_mixin.py:214-270:
        main weight = 1 - sum(loss_weights), the model's decoder + loss on every captured (output, length) pair) restated."""
        captured = {}
        e = self.encoder
        e.is_access_enabled = lambda guid=None: True
        e.access_cfg = {"interctc": {"capture_layers": list(apply_at_layers)}}
        e.interctc_capture_at_layers = None
        e.register_accessible_tensor = lambda name, tensor: captured.__setitem__(name, tensor)
        try:
            loss, logp, enc, enc_len, mel, mel_len = self.forward(audio, audio_len, tokens, token_len)
        finally:
            del e.is_access_enabled, e.register_accessible_tensor
            e.interctc_capture_at_layers = None
        out = {"final_loss": loss}
        total = loss * (1.0 - sum(loss_weights))
        for l, w in zip(apply_at_layers, loss_weights):
            x, n = captured[f"interctc/layer_output_{l}"], captured[f"interctc/layer_length_{l}"]
            lp = torch.log_softmax(self.decoder_layers(x).transpose(1, 2), dim=-1)
            inter = self.ctc(lp.transpose(1, 0), tokens.long(), n.long(), token_len.long()).mean()
            out[f"inter_ctc_loss_l{l}"] = inter
            out[f"layer_output_{l}"] = x
            total = total + inter * w
        out["loss"] = total
        return out, enc, enc_len
