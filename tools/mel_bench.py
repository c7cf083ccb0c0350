"""Front-end kernel alone: both variants of csrc/mel.hip at the headline shape (B = 32 x 20 s), HIP events, rotated buffers.
    python tools/mel_bench.py            -> us per call, GB/s of algorithmic traffic (4 B/sample read + 4 B per mel value written)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nemo_amd import ops
from nemo_amd._lib import lib
from nemo_amd.modules.audio_preprocessing import sparsify_filterbank
from oracle import conformer_ref as R  # filterbank / window generators only

dev = torch.device("cuda:0")
B, S = 32, 320000
fb = tuple(t.to(dev) for t in sparsify_filterbank(torch.from_numpy(R.mel_filterbank())))
win = R.hann_window_sym(400).to(dev)
sets = [(0.1 * torch.randn(B, S, device=dev), torch.full((B,), S, device=dev, dtype=torch.int64),
         torch.empty(B, 80, 1 + S // 160, device=dev)) for _ in range(6)]
bytes_ = B * S * 4 + B * 80 * (1 + S // 160) * 4
for dither in (0.0, 1e-5):
    for variant in (0, 1, 2, 1, 2):
        lib.mi355x_logmel_config(variant)
        for a, l, o_ in sets:
            ops.logmel(a, l, win, fb, 80, dither=dither, seed=7, out=o_)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 0
        e0.record()
        for _ in range(5):
            for a, l, o_ in sets:
                ops.logmel(a, l, win, fb, 80, dither=dither, seed=7, out=o_)
                n += 1
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(f"dither={dither:g} variant={variant}: {us:.1f} us per call, {bytes_ / us / 1e3:.0f} GB/s algorithmic")
lib.mi355x_logmel_config(2)
