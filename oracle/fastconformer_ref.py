"""TEST INFRASTRUCTURE ONLY -- the FastConformer encoder (cfg 4 of BASELINE.json) as a CPU restatement: the Conformer layers of
oracle/conformer_ref.py behind the 'dw_striding' x8 sub-sampling of oracle/squeezeformer_ref.py (three stride-2 stages,
256 channels in the recipe) with the depthwise kernel 9 of `examples/asr/conf/fastconformer/fast-conformer_transducer_bpe.yaml`.
Reference: `modules/conformer_encoder.py:593-759` with `subsampling='dw_striding'` (`subsampling.py:142-215`).
Pinned against the reference ConformerEncoder by tests/golden/ref_fastconformer_tiny.npz."""
from __future__ import annotations

import math

import torch

from . import conformer_ref as R
from . import squeezeformer_ref as SQ


def encoder_forward(P, cfg: R.ConformerCfg, mel, mel_len, bn_training: bool = False):
    x, enc_len = SQ.dw_striding_forward(P, mel, mel_len, cfg=cfg)  # (cfg.emulate_bf16: bf16 storage points)
    B, T, d = x.shape
    if cfg.xscaling:
        x = x * math.sqrt(d)
    pos_emb = R.rel_pos_table(T, d)
    valid = torch.arange(T).unsqueeze(0) < enc_len.unsqueeze(1)
    for i in range(cfg.n_layers):
        x = R.conformer_layer(P, f"layers.{i}.", cfg, x, pos_emb, valid, False, bn_training)
    return x.transpose(1, 2), enc_len
