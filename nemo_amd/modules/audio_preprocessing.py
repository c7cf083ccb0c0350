"""Drop-in for `nemo.collections.asr.modules.AudioToMelSpectrogramPreprocessor`
(nemo/collections/asr/modules/audio_preprocessing.py:111-330) and its `FilterbankFeatures` featurizer
(parts/preprocessing/features.py:246-502), executing on the fused HIP front-end (csrc/mel.hip).

Same constructor kwargs, typed I/O, attributes callers touch (`_sample_rate`, `featurizer.dither`, `featurizer.pad_to`,
`filter_banks`) and persistent buffers (`featurizer.window`, `featurizer.fb`) so a reference `.nemo` state-dict loads.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

from ..core import (AudioSignal, LengthsType, MelSpectrogramType, NeuralModule, NeuralType, SpectrogramType,
                    typecheck)

CONSTANT = 1e-5


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * 27.0 / np.log(6.4), f * 3.0 / 200.0)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((m - 15.0) * np.log(6.4) / 27.0), m * 200.0 / 3.0)


def slaney_mel_filterbank(sr, n_fft, n_mels, fmin=0.0, fmax=None, norm="slaney") -> np.ndarray:
    """= librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm) (third-party, librosa>=0.10.1; the reference
    calls it at features.py:338-344).  Slaney scale, triangular weights on rfftfreq, area normalisation."""
    fmax = sr / 2.0 if fmax is None else fmax
    edges = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    bins = np.fft.rfftfreq(n_fft, 1.0 / sr)
    fb = np.zeros((n_mels, bins.size))
    for i in range(n_mels):
        up = (bins - edges[i]) / (edges[i + 1] - edges[i])
        down = (edges[i + 2] - bins) / (edges[i + 2] - edges[i + 1])
        fb[i] = np.maximum(0.0, np.minimum(up, down))
        if norm == "slaney":
            fb[i] *= 2.0 / (edges[i + 2] - edges[i])
    return fb.astype(np.float32)


def sparsify_filterbank(fb: torch.Tensor):
    """fb [n_mels, n_bins] -> (start i32 [n_mels], len i32, offset i32, weights f32): per-row non-zero span.  Every row's weights
    start at a multiple of four floats and are zero-filled up to the next one: the front-end kernel reads them as 16-byte
    vectors without end-of-filter checks (csrc/mel.hip; a zero weight meets a finite power)."""
    fbn = fb.detach().float().cpu().numpy().reshape(fb.shape[-2], fb.shape[-1])
    starts, lens, offs, ws = [], [], [], []
    o = 0
    for row in fbn:
        nz = np.nonzero(row)[0]
        if nz.size == 0:
            s, n = 0, 0
        else:
            s, n = int(nz[0]), int(nz[-1] - nz[0] + 1)
        n4 = (n + 3) // 4 * 4
        starts.append(s); lens.append(n); offs.append(o)
        ws.append(np.concatenate([row[s: s + n], np.zeros(n4 - n, row.dtype)]))
        o += n4
    w = np.concatenate(ws) if o > 0 else np.zeros(4, np.float32)
    return (torch.tensor(starts, dtype=torch.int32), torch.tensor(lens, dtype=torch.int32),
            torch.tensor(offs, dtype=torch.int32), torch.from_numpy(w.astype(np.float32)))


class FilterbankFeatures(nn.Module):
    """features.py:246 -- same ctor arguments; `forward(x, seq_len)` -> (features [B, nfilt, T], seq_len)."""

    def __init__(self, sample_rate=16000, n_window_size=320, n_window_stride=160, window="hann", normalize="per_feature",
                 n_fft=None, preemph=0.97, nfilt=64, lowfreq=0, highfreq=None, log=True, log_zero_guard_type="add",
                 log_zero_guard_value=2 ** -24, dither=CONSTANT, pad_to=16, max_duration=16.7, frame_splicing=1,
                 exact_pad=False, pad_value=0, mag_power=2.0, use_grads=False, rng=None, nb_augmentation_prob=0.0,
                 nb_max_freq=4000, mel_norm="slaney", stft_exact_pad=False, stft_conv=False):
        super().__init__()
        if exact_pad and n_window_stride % 2 == 1:
            raise NotImplementedError(f"{self} received exact_pad == True, but hop_size was odd.")
        if (n_window_size is None or n_window_stride is None or not isinstance(n_window_size, int)
                or not isinstance(n_window_stride, int) or n_window_size <= 0 or n_window_stride <= 0):
            raise ValueError(f"{self} got an invalid value for either n_window_size or n_window_stride. "
                             f"Both must be positive ints.")
        if log_zero_guard_type not in ["add", "clamp"]:
            raise ValueError(f"{self} received {log_zero_guard_type} for the log_zero_guard_type parameter. "
                             f"It must be either 'add' or 'clamp'.")
        # what the fused HIP front-end implements (everything the Conformer-CTC configs use)
        unsupported = []
        if exact_pad: unsupported.append("exact_pad=True")
        if frame_splicing != 1: unsupported.append("frame_splicing>1")
        if mag_power != 2.0: unsupported.append("mag_power!=2")
        if not log or log_zero_guard_type != "add": unsupported.append("log=False / clamp guard")
        if nb_augmentation_prob > 0: unsupported.append("nb_augmentation_prob>0")
        if use_grads: unsupported.append("use_grads=True")
        # normalize_batch (features.py:58-112) leaves the features as they are for any type it does not know -- the streaming
        # recipes' `normalize: "NA"` is exactly that; "all_features" and fixed statistics are not implemented here
        if normalize == "all_features" or isinstance(normalize, dict): unsupported.append(f"normalize={normalize}")
        if window not in ("hann", "hamming", "blackman", "bartlett", "none", None): unsupported.append(f"window={window}")
        if unsupported:
            raise NotImplementedError("MI355X front-end does not implement: " + ", ".join(unsupported))
        self.log_zero_guard_value = log_zero_guard_value
        self.sample_rate = sample_rate
        self.win_length = n_window_size
        self.hop_length = n_window_stride
        self.n_fft = n_fft or 2 ** math.ceil(math.log2(self.win_length))
        if self.n_fft != 512:
            raise NotImplementedError("MI355X front-end is specialised for n_fft = 512")
        self.stft_pad_amount = None
        self.exact_pad = False
        windows = {"hann": torch.hann_window, "hamming": torch.hamming_window, "blackman": torch.blackman_window,
                   "bartlett": torch.bartlett_window, "none": None, None: None}
        fn = windows.get(window)
        win = fn(self.win_length, periodic=False) if fn else torch.ones(self.win_length)
        self.register_buffer("window", win)
        self.normalize = normalize
        self.log = log
        self.dither = dither
        self.frame_splicing = frame_splicing
        self.nfilt = nfilt
        self.preemph = preemph
        self.pad_to = pad_to
        highfreq = highfreq or sample_rate / 2
        fb = torch.tensor(slaney_mel_filterbank(sample_rate, self.n_fft, nfilt, lowfreq, highfreq, mel_norm)).unsqueeze(0)
        self.register_buffer("fb", fb)
        max_length = int(self.get_seq_len(torch.tensor(max_duration * sample_rate, dtype=torch.float)))
        max_pad = pad_to - (max_length % pad_to) if pad_to > 0 else 0
        self.max_length = max_length + max_pad
        self.pad_value = pad_value
        self.mag_power = mag_power
        self.log_zero_guard_type = log_zero_guard_type
        self._sparse = None
        self._sparse_key = None
        self._seed = 0

    def get_seq_len(self, seq_len):
        pad_amount = self.n_fft // 2 * 2
        return torch.floor_divide(seq_len + pad_amount - self.n_fft, self.hop_length).to(dtype=torch.long)

    @property
    def filter_banks(self):
        return self.fb

    def _fb_sparse(self):
        key = (self.fb.data_ptr(), self.fb._version, str(self.fb.device))
        if self._sparse is None or self._sparse_key != key:
            self._sparse = tuple(t.to(self.fb.device) for t in sparsify_filterbank(self.fb))
            self._sparse_key = key
        return self._sparse

    @torch.no_grad()
    def forward(self, x, seq_len, linear_spec=False, out_dtype=torch.float32):
        from .. import ops

        if linear_spec:
            raise NotImplementedError("linear_spec=True is not on the Conformer-CTC path")
        seq_len_unfixed = self.get_seq_len(seq_len)
        out_len = torch.where(seq_len == 0, torch.zeros_like(seq_len_unfixed), seq_len_unfixed)
        host = getattr(seq_len, "host_lengths", None)
        if host is not None:
            # the caller (input pipeline) knows the sample counts on the host: the frame counts follow by the same formula, so the
            # encoder can size a packed launch sequence (ConformerEncoder._packing_plan) without reading anything back
            h = torch.as_tensor(host, dtype=torch.int64)
            hu = self.get_seq_len(h)
            out_len.host_lengths = torch.where(h == 0, torch.zeros_like(hu), hu)
        x = x.contiguous()
        dither = self.dither if (self.training and self.dither > 0) else 0.0
        self._seed = (self._seed + 1) & 0x7FFFFFFF
        lg = self.log_zero_guard_value
        if isinstance(lg, str):
            lg = torch.finfo(torch.float32).tiny if lg == "tiny" else torch.finfo(torch.float32).eps
        raw = ops.logmel(x, seq_len.to(torch.int64), self.window.float(), self._fb_sparse(), self.nfilt,
                         hop=self.hop_length, n_fft=self.n_fft, preemph=self.preemph, dither=dither, seed=self._seed,
                         log_guard=float(lg))
        feat = ops.feat_normalize(raw, out_len, normalize=(self.normalize == "per_feature"), pad_value=float(self.pad_value),
                                  out_dtype=out_dtype)
        pad_to = self.pad_to
        if pad_to == "max":
            feat = nn.functional.pad(feat, (0, self.max_length - feat.size(-1)), value=self.pad_value)
        elif pad_to > 0:
            pad_amt = feat.size(-1) % pad_to
            if pad_amt != 0:
                feat = nn.functional.pad(feat, (0, pad_to - pad_amt), value=self.pad_value)
        return feat, out_len


class AudioToMelSpectrogramPreprocessor(NeuralModule):
    """audio_preprocessing.py:111 -- identical kwargs (:214-244) and typed I/O (:188-212)."""

    @property
    def input_types(self):
        return OrderedDict({"input_signal": NeuralType(("B", "T"), AudioSignal(freq=self._sample_rate)),
                            "length": NeuralType(tuple("B"), LengthsType())})

    @property
    def output_types(self):
        return OrderedDict({"processed_signal": NeuralType(("B", "D", "T"), MelSpectrogramType()),
                            "processed_length": NeuralType(tuple("B"), LengthsType())})

    def __init__(self, sample_rate=16000, window_size=0.02, window_stride=0.01, n_window_size=None, n_window_stride=None,
                 window="hann", normalize="per_feature", n_fft=None, preemph=0.97, features=64, lowfreq=0, highfreq=None,
                 log=True, log_zero_guard_type="add", log_zero_guard_value=2 ** -24, dither=1e-5, pad_to=16,
                 frame_splicing=1, exact_pad=False, pad_value=0, mag_power=2.0, rng=None, nb_augmentation_prob=0.0,
                 nb_max_freq=4000, use_torchaudio: bool = False, mel_norm="slaney", stft_exact_pad=False, stft_conv=False):
        super().__init__()
        self._sample_rate = sample_rate
        if window_size and n_window_size:
            raise ValueError(f"{self} received both window_size and n_window_size. Only one should be specified.")
        if window_stride and n_window_stride:
            raise ValueError(f"{self} received both window_stride and n_window_stride. Only one should be specified.")
        if window_size:
            n_window_size = int(window_size * self._sample_rate)
        if window_stride:
            n_window_stride = int(window_stride * self._sample_rate)
        if use_torchaudio:
            raise NotImplementedError("use_torchaudio=True is not supported by the MI355X front-end")
        self.win_length = n_window_size
        self.hop_length = n_window_stride
        self.register_buffer("dtype_sentinel_tensor", torch.tensor((), dtype=torch.float32), persistent=False)
        self.featurizer = FilterbankFeatures(
            sample_rate=self._sample_rate, n_window_size=n_window_size, n_window_stride=n_window_stride, window=window,
            normalize=normalize, n_fft=n_fft, preemph=preemph, nfilt=features, lowfreq=lowfreq, highfreq=highfreq, log=log,
            log_zero_guard_type=log_zero_guard_type, log_zero_guard_value=log_zero_guard_value, dither=dither, pad_to=pad_to,
            frame_splicing=frame_splicing, exact_pad=exact_pad, pad_value=pad_value, mag_power=mag_power, rng=rng,
            nb_augmentation_prob=nb_augmentation_prob, nb_max_freq=nb_max_freq, mel_norm=mel_norm,
            stft_exact_pad=stft_exact_pad, stft_conv=stft_conv)

    @typecheck()
    @torch.no_grad()
    def forward(self, input_signal, length):
        # the reference casts a non-fp32 input to fp32 with a warning (audio_preprocessing.py:95-101)
        feats, out_len = self.featurizer(input_signal.to(torch.float32), length)
        return feats.to(self.dtype_sentinel_tensor.dtype), out_len

    def get_features(self, input_signal, length):
        return self.featurizer(input_signal, length)

    @property
    def filter_banks(self):
        return self.featurizer.filter_banks

    def input_example(self, max_batch: int = 8, max_dim: int = 32000, min_length: int = 200):
        dev = self.filter_banks.device
        signals = torch.randn(size=[max_batch, max_dim], device=dev)
        lengths = torch.randint(low=min_length, high=max_dim, size=[max_batch], device=dev)
        lengths[0] = max_dim
        return signals, lengths


class SpectrogramAugmentation(NeuralModule):
    """audio_preprocessing.py:443-553 -- same kwargs, typed I/O and random streams.

    The mask PARAMETERS are drawn exactly as the reference draws them (vectorised mode: four `torch.rand((B, n))` calls on
    the spectrogram's device, time masks first -- spectr_augment.py:134-215; legacy mode and SpecCutout: the python
    `random.Random` stream -- :99-132, :245-261), so a seeded run masks the same cells.  The masks are then applied by ONE
    `mi355x_fill_rects` launch that writes only the masked cells (the reference builds a full boolean mask and runs
    `masked_fill` over the whole spectrogram twice).  The Numba kernel option is accepted and ignored (same result).
    """

    @property
    def input_types(self):
        return OrderedDict({"input_spec": NeuralType(("B", "D", "T"), SpectrogramType()),
                            "length": NeuralType(tuple("B"), LengthsType())})

    @property
    def output_types(self):
        return OrderedDict({"augmented_spec": NeuralType(("B", "D", "T"), SpectrogramType())})

    def __init__(self, freq_masks=0, time_masks=0, freq_width=10, time_width=10, rect_masks=0, rect_time=5, rect_freq=20,
                 rng=None, mask_value=0.0, use_vectorized_spec_augment: bool = True, use_numba_spec_augment: bool = False):
        super().__init__()
        import random as _random
        self._rng = _random.Random() if rng is None else rng
        self.freq_masks, self.time_masks = int(freq_masks), int(time_masks)
        self.freq_width, self.time_width = freq_width, time_width
        self.rect_masks, self.rect_time, self.rect_freq = int(rect_masks), rect_time, rect_freq
        self.mask_value = float(mask_value)
        self.use_vectorized_code = bool(use_vectorized_spec_augment)
        self.fused_rects = os.environ.get("MI355X_SPECAUG_FUSED", "1") != "0"   # mask parameters in one launch (device tensors)
        if not isinstance(time_width, int) and (time_width > 1.0 or time_width < 0.0):
            raise ValueError("If `time_width` is a float value, must be in range [0, 1]")

    # ---- mask parameters (device-agnostic torch / python code: runs on CPU tensors in the tests)
    def _vectorized_rects(self, B, F, T, length, device):
        if self.fused_rects and torch.device(device).type == "cuda" and isinstance(self.freq_width, int) \
                and (self.time_masks + self.freq_masks) > 0 and length.dtype == torch.int64:
            # the reference's four draws, in its order (time width, time start, frequency width, frequency start: the generator
            # stream is the reference's), then ONE launch for the ~40 tensor ops below (mi355x_specaug_rects: same f32 arithmetic)
            from .. import ops
            u = [torch.rand((B, n), device=device, dtype=torch.float32) for n in (self.time_masks, self.time_masks,
                                                                                   self.freq_masks, self.freq_masks)]
            rects = torch.empty((B * (self.time_masks + self.freq_masks), 5), dtype=torch.int32, device=device)
            return ops.specaug_rects(u[0], u[1], u[2], u[3], length.contiguous(), rects, B, self.time_masks, self.freq_masks, F, T,
                                     self.time_width, self.freq_width)
        rows = []
        bidx = torch.arange(B, device=device).unsqueeze(1)
        for num, width, is_time in ((self.time_masks, self.time_width, True), (self.freq_masks, self.freq_width, False)):
            axis_length = T if is_time else F
            if is_time and isinstance(width, float):
                width = torch.clamp(width * length, max=axis_length).unsqueeze(1)
            mask_width = (torch.rand((B, num), device=device, dtype=torch.float32) * width).long()
            mask_start = torch.rand((B, num), device=device, dtype=torch.float32)
            mask_start = (mask_start * ((length.unsqueeze(1) if is_time else axis_length) - mask_width)).long()
            mask_end = mask_start + mask_width
            z = torch.zeros_like(mask_start)
            if num == 0:
                continue
            if is_time:
                rows.append(torch.stack([bidx.expand_as(z), z, z + F, mask_start, mask_end], -1).reshape(-1, 5))
            else:
                rows.append(torch.stack([bidx.expand_as(z), mask_start, mask_end, z, z + T], -1).reshape(-1, 5))
        return torch.cat(rows, 0) if rows else torch.zeros(0, 5, dtype=torch.long, device=device)

    def _legacy_rects(self, B, F, T, length):
        lengths = [int(v) for v in length.tolist()]  # host sync, as in the reference (:101)
        rows = []
        for idx in range(B):
            for _ in range(self.freq_masks):
                start = self._rng.randint(0, F - self.freq_width)
                width = self._rng.randint(0, self.freq_width)
                rows.append((idx, start, start + width, 0, T))
            tw = max(1, int(lengths[idx] * self.time_width)) if isinstance(self.time_width, float) else self.time_width
            upper = max(1, lengths[idx] - tw)
            for _ in range(self.time_masks):
                start = self._rng.randint(0, upper)
                width = self._rng.randint(0, tw)
                rows.append((idx, 0, F, start, start + width))
        return torch.tensor(rows, dtype=torch.long).reshape(-1, 5)

    def _cutout_rects(self, B, F, T):
        rows = []
        for idx in range(B):
            for _ in range(self.rect_masks):
                x = self._rng.randint(0, F - self.rect_freq)
                y = self._rng.randint(0, T - self.rect_time)
                w_x = self._rng.randint(0, self.rect_freq)
                w_y = self._rng.randint(0, self.rect_time)
                rows.append((idx, x, x + w_x, y, y + w_y))
        return torch.tensor(rows, dtype=torch.long).reshape(-1, 5)

    def mask_rects(self, B, F, T, length, device):
        """[(rects int64 [n,5], value)] in application order: cut-out rectangles (zeros), then SpecAugment (mask_value)"""
        out = []
        if self.rect_masks > 0:
            out.append((self._cutout_rects(B, F, T).to(device), 0.0))
        if self.freq_masks + self.time_masks > 0:
            r = self._vectorized_rects(B, F, T, length.to(device), device) if self.use_vectorized_code \
                else self._legacy_rects(B, F, T, length).to(device)
            out.append((r, self.mask_value))
        return out

    @typecheck()
    @torch.no_grad()
    def forward(self, input_spec, length):
        B, F, T = input_spec.shape
        groups = self.mask_rects(B, F, T, length, input_spec.device)
        if not groups:
            return input_spec
        out = input_spec.to(torch.float32).clone() if self.rect_masks == 0 else input_spec.to(torch.float32)
        out = out.contiguous()
        from .. import ops
        for rects, value in groups:  # SpecCutout writes in place in the reference, SpecAugment returns a new tensor
            ops.fill_rects(out, rects.to(torch.int32).contiguous(), value)
        return out.to(input_spec.dtype)

