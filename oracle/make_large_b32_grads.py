"""TEST INFRASTRUCTURE -- one-off generator of tests/golden/oracle_large_b32_grads.npz: the per-tensor GRADIENTS of BASELINE.json
configs[1] at the batch the benchmark times (Conformer-CTC-Large, B = 32 x 20 s, R.synthetic_batch(32, 20.0, vocab=128,
seed=1234), weights R.init_params(ConformerCfg.large, seed=0)) by the CPU oracle (oracle/conformer_ref.py, pinned to the
reference's own files by tests/test_oracle_pinning.py), in fp32 and with bf16 rounding emulated at the HIP path's storage points.
121.5 M gradient values per run cannot be committed: per tensor the file keeps the fp32 run's L2 norm, max |.| and K = 16
pseudo-random +-1 projections (R.grad_projections: they give back the L2 distance of any other run to it), and the EXACT relative
L2 distance of the bf16-emulating run to the fp32 run -- the yardstick of tests/test_baseline_configs_gpu.py's derived tolerance.
The layers run under activation checkpointing (the fp32 autograd graph of this batch does not fit the container's 62 GB
otherwise); gradients are unchanged by that.  Minutes per run:

    python -m oracle.make_large_b32_grads
"""
import dataclasses
import math
import os
import time

import numpy as np
import torch
from torch.utils.checkpoint import checkpoint

from . import conformer_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = 16


def run(P, cfg, batch):
    audio, alen, tok, tl = batch
    Pd = {k: v.detach().clone() for k, v in P.items()}
    keys = R.trainable_keys(Pd)
    for k in keys:
        Pd[k].requires_grad_(True)
    with torch.no_grad():
        mel, mel_len = R.log_mel_features(audio, alen, n_mels=cfg.feat_in)
    # encoder_forward of the oracle, layer by layer under checkpointing (same functions, same order: conformer_ref.py:342-355)
    pfx = "encoder."
    x, enc_len = checkpoint(lambda m: R.subsampling_forward(Pd, cfg, m, mel_len, pfx + "pre_encode."), mel.requires_grad_(True),
                            use_reentrant=False)
    B, T, d = x.shape
    if cfg.xscaling:
        x = x * math.sqrt(d)
    pos_emb = R.rel_pos_table(T, d).to(x.dtype)
    valid = torch.arange(T).unsqueeze(0) < enc_len.unsqueeze(1)
    for i in range(cfg.n_layers):
        x = checkpoint(lambda xx, i=i: R.conformer_layer(Pd, f"{pfx}layers.{i}.", cfg, xx, pos_emb, valid, False, True, None), x,
                       use_reentrant=False)
    logp = R.decoder_forward(Pd, x.transpose(1, 2), "decoder.decoder_layers.0.", cfg)
    loss, per = R.ctc_loss_mean_batch(logp, tok, enc_len, tl, cfg.vocab)
    loss.backward()
    return float(loss.detach()), {k: Pd[k].grad.detach() for k in keys}


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = R.ConformerCfg.large(vocab=128, dropout=0.0, dropout_att=0.0, dropout_pre_encoder=0.0)
    P = R.init_params(cfg, seed=0)
    batch = R.synthetic_batch(32, 20.0, vocab=128, seed=1234)
    t0 = time.time()
    l32, g32 = run(P, cfg, batch)
    print("fp32", l32, round(time.time() - t0, 1), "s", flush=True)
    t0 = time.time()
    lemu, gemu = run(P, dataclasses.replace(cfg, emulate_bf16=True), batch)
    print("bf16_emulated", lemu, round(time.time() - t0, 1), "s", flush=True)
    names = sorted(g32)
    norm = np.array([g32[n].double().norm().item() for n in names])
    amax = np.array([g32[n].double().abs().max().item() for n in names])
    e_emu = np.array([(gemu[n].double() - g32[n].double()).norm().item() / max(g32[n].double().norm().item(), 1e-300) for n in names])
    proj = np.stack([R.grad_projections(n, g32[n], K).numpy() for n in names])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_large_b32_grads.npz"), names=np.array(names), norm=norm,
                        amax=amax, e_emu=e_emu, proj=proj, loss_fp32=l32, loss_bf16_emulated=lemu, K=K,
                        config="Conformer-CTC-Large, B=32x20s, R.synthetic_batch(32, 20.0, vocab=128, seed=1234), "
                               "R.init_params(large, seed=0), train-mode BatchNorm statistics, no dropout / dither / SpecAugment; torch "
                               + torch.__version__)
    print("tensors", len(names), "numel", sum(g32[n].numel() for n in names), flush=True)


if __name__ == "__main__":
    main()
