"""RNN-T loss + gradient timing at FastConformer-Transducer joint sizes (20 s / 8x sub-sampling = 250 frames, 60 tokens, V = 1024)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemo_amd.modules import RNNTLoss

dev = "cuda"
for B in (4, 16):
    T, U1, V1 = 250, 61, 1025
    a = torch.randn(B, T, U1, V1, device=dev, requires_grad=True)
    lab = torch.randint(0, 1024, (B, U1 - 1), device=dev)
    lens = torch.full((B,), T, device=dev); ll = torch.full((B,), U1 - 1, device=dev)
    f = RNNTLoss(blank=1024, reduction="mean")
    for i in range(3):
        c = f(a, lab, lens, ll)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(10):
        c = f(a, lab, lens, ll)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gb = 2 * a.numel() * 4 / 1e9  # compulsory: read the logits once, write the gradients once
    print(f"rnnt B={B} T={T} U1={U1} V1={V1}: {ms:.3f} ms per loss+grad, {gb / ms:.2f} TB/s of compulsory traffic ({gb:.2f} GB)")
