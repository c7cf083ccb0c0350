"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's Squeezeformer encoder, plain
torch on a state-dict `P` whose keys are the reference module's own.  Prepared ahead of the HIP path: the oracle comes first.

Follows `nemo/collections/asr/modules/squeezeformer_encoder.py:300-372` (forward_for_export: pre_encode -> pos_enc with
xscale -> pre_ln -> layers with the time-reduction / recovery detour), `parts/submodules/squeezeformer_modules.py:30-203`
(`ScaleBiasLayer`; `SqueezeformerLayer.forward`: MHA -> FFN1 -> Conv -> FFN2, each `LN(residual + f(x*scale + bias))`),
`conformer_modules.py:236-350` with `pointwise_activation='swish'` (Swish instead of GLU, so depthwise conv / BatchNorm /
pointwise_conv2 run on 2*d_model channels), `subsampling.py:142-215, 385-436, 725-759` ('dw_striding': Conv2d(1->C,3,s2) ReLU
[depthwise Conv2d(C,3,s2,groups=C) + pointwise Conv2d(C->C,1)] ReLU under MaskedConvSequential, Linear) and `:589-646`
(`TimeReductionModule`: masked depthwise Conv1d(k=5,s=2,pad=3) + pointwise, masks strided by 2, output padded to ceil(T/2)).
Pinned against the reference classes themselves (run here through oracle/ref_shim.py) by tests/golden/ref_squeezeformer_tiny.npz."""
from __future__ import annotations

import dataclasses
import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F
from torch import Tensor

from . import conformer_ref as R


@dataclasses.dataclass
class SqueezeformerCfg:
    feat_in: int = 80
    d_model: int = 144
    n_heads: int = 4
    n_layers: int = 16
    ff_expansion: int = 4
    conv_kernel: int = 31
    xscaling: bool = True
    time_reduce_idx: Optional[int] = None
    time_recovery_idx: Optional[int] = None
    dropout_att: float = 0.0

    @property
    def d_k(self):
        return self.d_model // self.n_heads


def dw_striding_forward(P: Dict[str, Tensor], mel: Tensor, mel_len: Tensor, pfx="pre_encode."):
    """mel [B, F, T] -> ([B, T', d], lengths): every layer of the stack sees a masked input (MaskedConvSequential)"""
    B = mel.shape[0]
    x = mel.transpose(1, 2).unsqueeze(1)
    cur = mel_len.clone().float()

    def mask(t, n):
        return (torch.arange(t.shape[2]).unsqueeze(0) < n.long().unsqueeze(1)).to(t.dtype).view(B, 1, -1, 1)

    C = P[pfx + "conv.0.weight"].shape[0]
    stack = [("conv", "conv.0", 2, 1, 1), ("relu",)]
    idx = 2
    while f"{pfx}conv.{idx}.weight" in P:  # one (depthwise s2, pointwise, ReLU) group per further factor of 2 (x4: one, x8: two)
        stack += [("conv", f"conv.{idx}", 2, 1, C), ("conv", f"conv.{idx + 1}", 1, 0, 1), ("relu",)]
        idx += 3
    m = mask(x, cur)
    for layer in stack:
        x = x * m
        if layer[0] == "relu":
            x = torch.relu(x)
            continue
        _, name, stride, pad, groups = layer
        x = F.conv2d(x, P[pfx + name + ".weight"], P[pfx + name + ".bias"], stride=stride, padding=pad, groups=groups)
        if stride != 1:
            cur = torch.div(cur + 2 * pad - 3, stride, rounding_mode="floor") + 1  # calculate_conv_output_size
            m = mask(x, cur)
    x = x * m
    b, c, t, f = x.shape
    x = F.linear(x.transpose(1, 2).reshape(b, t, c * f), P[pfx + "out.weight"], P[pfx + "out.bias"])
    return x, cur.long()


def swish_conv_module(P, pfx, x: Tensor, valid: Tensor, kernel: int, bn_training: bool):
    """ConformerConvolution(pointwise_activation='swish'): all of the depthwise / BatchNorm work on 2*d channels"""
    h = F.linear(x, P[pfx + "pointwise_conv1.weight"].squeeze(-1), P[pfx + "pointwise_conv1.bias"])  # [B,T,2d]
    h = h * torch.sigmoid(h)
    h = h * valid.unsqueeze(-1).to(h.dtype)
    c2 = h.shape[-1]
    pad = (kernel - 1) // 2
    c = F.conv1d(F.pad(h.transpose(1, 2), (pad, pad)), P[pfx + "depthwise_conv.weight"], P[pfx + "depthwise_conv.bias"],
                 groups=c2)
    if bn_training:
        mean, var = c.mean(dim=(0, 2)), c.var(dim=(0, 2), unbiased=False)
    else:
        mean, var = P[pfx + "batch_norm.running_mean"], P[pfx + "batch_norm.running_var"]
    c = (c - mean.view(1, c2, 1)) * torch.rsqrt(var.view(1, c2, 1) + 1e-5)
    c = c * P[pfx + "batch_norm.weight"].view(1, c2, 1) + P[pfx + "batch_norm.bias"].view(1, c2, 1)
    c = c * torch.sigmoid(c)
    return F.linear(c.transpose(1, 2), P[pfx + "pointwise_conv2.weight"].squeeze(-1), P[pfx + "pointwise_conv2.bias"])


def _sb(P, pfx, x):
    return x * P[pfx + "scale"] + P[pfx + "bias"]


def squeezeformer_layer(P, pfx, cfg: SqueezeformerCfg, x, pos_emb, valid, bn_training):
    acfg = R.ConformerCfg(d_model=cfg.d_model, n_heads=cfg.n_heads, n_layers=1, dropout=0, dropout_att=cfg.dropout_att,
                          dropout_pre_encoder=0)
    x = R._ln(P, pfx + "norm_self_att.", x + R.rel_pos_attention(P, pfx + "self_attn.", acfg, _sb(P, pfx + "self_attn_scale.", x),
                                                               pos_emb, valid, False))
    x = R._ln(P, pfx + "norm_feed_forward1.", x + R.feed_forward(P, pfx + "feed_forward1.", acfg,
                                                                _sb(P, pfx + "feed_forward1_scale.", x), False))
    x = R._ln(P, pfx + "norm_conv.", x + swish_conv_module(P, pfx + "conv.", _sb(P, pfx + "conv_scale.", x), valid,
                                                          cfg.conv_kernel, bn_training))
    x = R._ln(P, pfx + "norm_feed_forward2.", x + R.feed_forward(P, pfx + "feed_forward2.", acfg,
                                                                _sb(P, pfx + "feed_forward2_scale.", x), False))
    return x


def time_reduction(P, pfx, x: Tensor, valid: Tensor):
    """x [B,T,d], valid [B,T] -> ([B, ceil(T/2), d], valid[:, ::2])"""
    h = (x * valid.unsqueeze(-1).to(x.dtype)).transpose(1, 2)
    d = h.shape[1]
    h = F.conv1d(h, P[pfx + "dw_conv.weight"], P[pfx + "dw_conv.bias"], stride=2, padding=3, groups=d)
    h = F.conv1d(h, P[pfx + "pw_conv.weight"], P[pfx + "pw_conv.bias"]).transpose(1, 2)
    v2 = valid[:, ::2]
    return F.pad(h, (0, 0, 0, v2.shape[1] - h.shape[1])), v2


def encoder_forward(P: Dict[str, Tensor], cfg: SqueezeformerCfg, mel: Tensor, mel_len: Tensor, bn_training: bool = False):
    """-> (encoded [B, d, T'], lengths [B]); dropout = 0 (parity configuration)"""
    x, enc_len = dw_striding_forward(P, mel, mel_len)
    B, T, d = x.shape
    if cfg.xscaling:
        x = x * math.sqrt(d)
    pos_emb = R.rel_pos_table(T, d).to(x.dtype)
    valid = torch.arange(T).unsqueeze(0) < enc_len.unsqueeze(1)
    x = R._ln(P, "pre_ln.", x)
    rec_idx = cfg.time_recovery_idx if cfg.time_recovery_idx is not None else cfg.n_layers - 1
    cache = None
    for i in range(cfg.n_layers):
        if cfg.time_reduce_idx is not None and i == cfg.time_reduce_idx:
            cache = (x, valid, pos_emb)
            x, valid = time_reduction(P, "time_reduce_layer.", x, valid)
            pos_emb = R.rel_pos_table(x.shape[1], d).to(x.dtype)
        if cfg.time_reduce_idx is not None and i == rec_idx:
            x0, valid, pos_emb = cache
            x = torch.repeat_interleave(x, repeats=2, dim=1)[:, : x0.shape[1]]
            x = x0 + F.linear(x, P["time_recovery_layer.weight"], P["time_recovery_layer.bias"])
        x = squeezeformer_layer(P, f"layers.{i}.", cfg, x, pos_emb, valid, bn_training)
    return x.transpose(1, 2), enc_len
