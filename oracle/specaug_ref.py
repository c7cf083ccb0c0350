"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's SpectrogramAugmentation path.

Follows nemo/collections/asr/parts/submodules/spectr_augment.py:
  * `vectorized_rects`  = SpecAugment._forward_vectorized / _apply_masks (:134-215): time masks first, then frequency masks;
    per axis  width = (rand(B, n) * max_width).long(),  start = (rand(B, n) * (extent - width)).long()  with the
    default torch generator of the spectrogram's device; a float time_width means max_width = clamp(w * length, max=T).
  * `legacy_rects`      = SpecAugment._forward_legacy (:99-132): python `random.Random` stream, per utterance 2 draws per
    frequency mask then 2 per time mask; adaptive width max(1, int(len * w)).
  * `cutout_rects`      = SpecCutout.forward (:245-261): 4 draws per rectangle.
  * `apply_rects`       = the masked_fill / slice assignment.
Rectangles are rows (b, f0, f1, t0, t1) with half-open ranges, clipped on application -- the format of `mi355x_fill_rects`.
Pinned against the reference classes (loaded through oracle/ref_shim.py) by oracle/make_golden.py ->
tests/golden/ref_specaug.npz and tests/test_oracle_pinning.py.  Nothing under nemo_amd/ imports this file.
"""
from __future__ import annotations

import numpy as np
import torch


def vectorized_rects(B, F, T, length, freq_masks, time_masks, freq_width, time_width, device="cpu"):
    """returns int64 tensor [B*(time_masks+freq_masks), 5]; consumes torch's default generator exactly like the reference"""
    length = length.to(device)
    rows = []
    bidx = torch.arange(B, device=device).unsqueeze(1)

    def axis_masks(num, width, extent_is_time):
        axis_length = T if extent_is_time else F
        if extent_is_time and isinstance(width, float):
            width = torch.clamp(width * length, max=axis_length).unsqueeze(1)
        mask_width = (torch.rand((B, num), device=device, dtype=torch.float32) * width).long()
        mask_start = torch.rand((B, num), device=device, dtype=torch.float32)
        if extent_is_time:
            mask_start = mask_start * (length.unsqueeze(1) - mask_width)
        else:
            mask_start = mask_start * (axis_length - mask_width)
        mask_start = mask_start.long()
        return mask_start, mask_start + mask_width

    if time_masks > 0:  # _forward_vectorized always calls _apply_masks for both axes; num_masks = 0 draws empty tensors
        s, e = axis_masks(time_masks, time_width, True)
        z = torch.zeros_like(s)
        rows.append(torch.stack([bidx.expand_as(s), z, z + F, s, e], -1).reshape(-1, 5))
    else:
        axis_masks(0, time_width, True)
    if freq_masks > 0:
        s, e = axis_masks(freq_masks, freq_width, False)
        z = torch.zeros_like(s)
        rows.append(torch.stack([bidx.expand_as(s), s, e, z, z + T], -1).reshape(-1, 5))
    else:
        axis_masks(0, freq_width, False)
    return torch.cat(rows, 0) if rows else torch.zeros(0, 5, dtype=torch.long, device=device)


def legacy_rects(rng, B, F, T, length, freq_masks, time_masks, freq_width, time_width):
    lengths = [int(v) for v in length.tolist()]
    rows = []
    freq_start_upper_bound = F - freq_width
    for idx in range(B):
        for _ in range(freq_masks):
            start = rng.randint(0, freq_start_upper_bound)
            width = rng.randint(0, freq_width)
            rows.append((idx, start, start + width, 0, T))
        if isinstance(time_width, float):
            time_max_width = max(1, int(lengths[idx] * time_width))
        else:
            time_max_width = time_width
        time_start_upper_bound = max(1, lengths[idx] - time_max_width)
        for _ in range(time_masks):
            start = rng.randint(0, time_start_upper_bound)
            width = rng.randint(0, time_max_width)
            rows.append((idx, 0, F, start, start + width))
    return torch.tensor(rows, dtype=torch.long).reshape(-1, 5)


def cutout_rects(rng, B, F, T, rect_masks, rect_time, rect_freq):
    rows = []
    for idx in range(B):
        for _ in range(rect_masks):
            rect_x = rng.randint(0, F - rect_freq)
            rect_y = rng.randint(0, T - rect_time)
            w_x = rng.randint(0, rect_freq)
            w_y = rng.randint(0, rect_time)
            rows.append((idx, rect_x, rect_x + w_x, rect_y, rect_y + w_y))
    return torch.tensor(rows, dtype=torch.long).reshape(-1, 5)


def apply_rects(spec, rects, value=0.0):
    out = spec.clone()
    B, F, T = out.shape
    for b, f0, f1, t0, t1 in np.asarray(rects.cpu()).tolist():
        out[b, max(f0, 0):min(f1, F), max(t0, 0):min(t1, T)] = value
    return out
