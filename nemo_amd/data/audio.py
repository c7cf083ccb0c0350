"""Waveform loading for the map-style datasets: the `WaveformFeaturizer.process` -> `AudioSegment.from_file` slice
(`parts/preprocessing/features.py:189-222`, `segment.py:290-395`) that the Conformer-CTC recipes exercise: RIFF/WAVE PCM
read with `offset` / `duration` given in seconds (`seek(int(offset * sr))`, `read(int(duration * sr))`), integer PCM scaled
to float32 in [-1, 1) the way libsndfile's float read does (x / 2^(bits-1)), optional down-mix (`channel_selector`:
'average' or a channel index, segment.py:68-125).

The reference decodes through soundfile / pydub and resamples with librosa; neither is part of this image, so the
container formats are limited to what the standard library reads (PCM WAV, 8/16/24/32 bit) plus raw `.npy` float arrays,
and a file whose rate differs from `sample_rate` is an error instead of a silent resample."""
from __future__ import annotations

import wave
from typing import Optional, Union

import numpy as np
import torch


def _pcm_to_float(raw: bytes, width: int, int_values: bool) -> np.ndarray:
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.int32)
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4")
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        x = (b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16))
        x = np.where(x & 0x800000, x - (1 << 24), x)
    elif width == 1:
        x = np.frombuffer(raw, dtype=np.uint8).astype(np.int32) - 128
    else:
        raise ValueError(f"unsupported PCM sample width {width}")
    if int_values:  # soundfile dtype='int32' left-justifies, AudioSegment then scales by 2^-31 (segment.py:166-176)
        return (x.astype(np.float64) * (1 << (32 - 8 * width)) / float(1 << 31)).astype(np.float32)
    return (x.astype(np.float32) / np.float32(1 << (8 * width - 1))).astype(np.float32)


def load_audio(path: str, sample_rate: int, offset: float = 0.0, duration: float = 0.0, int_values: bool = False,
               channel_selector: Optional[Union[int, str]] = None) -> torch.Tensor:
    if path.endswith(".npy"):
        x = np.load(path, mmap_mode="r")
        sr = sample_rate
        start = int(offset * sr) if offset and offset > 0 else 0
        stop = start + int(duration * sr) if duration and duration > 0 else x.shape[0]
        x = np.asarray(x[start:stop], dtype=np.float32)
    else:
        with wave.open(path, "rb") as f:
            sr, nch, width = f.getframerate(), f.getnchannels(), f.getsampwidth()
            if offset is not None and offset > 0:
                f.setpos(min(int(offset * sr), f.getnframes()))
            n = int(duration * sr) if duration is not None and duration > 0 else f.getnframes()
            raw = f.readframes(n)
        x = _pcm_to_float(raw, width, int_values)
        if nch > 1:
            x = x.reshape(-1, nch)
    if x.ndim == 2:
        if channel_selector == "average":
            x = x.mean(axis=-1)
        elif isinstance(channel_selector, int):
            if channel_selector >= x.shape[-1]:
                raise ValueError(f"Cannot select channel {channel_selector} from a signal with {x.shape[-1]} channels.")
            x = x[..., channel_selector]
        else:
            raise ValueError(f"{path}: {x.shape[-1]} channels; set channel_selector ('average' or an index)")
    if sr != sample_rate:
        raise ValueError(f"{path}: sample rate {sr} != {sample_rate}; resample the corpus offline (no resampler in this image)")
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
