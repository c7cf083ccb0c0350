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
    # round to bf16 wherever the HIP bf16 path STORES bf16 (GEMM operand images of the weights, activations between kernels,
    # and the matching gradients): the rounding-only error of this run against the plain fp32 run is what the bf16 parity
    # tests derive their per-tensor tolerance from (same method as ConformerCfg.emulate_bf16)
    emulate_bf16: bool = False

    @property
    def d_k(self):
        return self.d_model // self.n_heads


def dw_striding_forward(P: Dict[str, Tensor], mel: Tensor, mel_len: Tensor, pfx="pre_encode.", cfg=None):
    """mel [B, F, T] -> ([B, T', d], lengths): every layer of the stack sees a masked input (MaskedConvSequential).
    bf16 storage points of the HIP path (cfg.emulate_bf16): the ReLU outputs (conv1 and every pointwise stage), the depthwise
    outputs, the pointwise / output-Linear weight images; conv1 and the depthwise kernels read fp32 weights."""
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
            x = R._q(torch.relu(x) * m, cfg)  # (the HIP epilogue masks and rounds in one store; relu(x*m)*m == relu(x*m))
            continue
        _, name, stride, pad, groups = layer
        w = P[pfx + name + ".weight"]
        pointwise = stride == 1
        causal = (not pointwise) and getattr(cfg, "causal_downsampling", False)
        if causal:  # CausalConv2D (causal_convs.py:24-72): pad (k - 1, stride - 1) = (2, 1) on time and frequency, conv without padding
            x = F.conv2d(F.pad(x, (2, 1, 2, 1)), w, P[pfx + name + ".bias"], stride=stride, groups=groups)
        else:
            x = F.conv2d(x, R._qw(w, cfg) if pointwise else w, P[pfx + name + ".bias"], stride=stride, padding=pad, groups=groups)
        if groups > 1:
            x = R._q(x, cfg)  # the depthwise output is the bf16 A operand of the pointwise GEMM
        if stride != 1:
            cur = torch.div(cur + (3 if causal else 2 * pad) - 3, stride, rounding_mode="floor") + 1  # calculate_conv_output_size
            m = mask(x, cur)
    x = x * m
    b, c, t, f = x.shape
    x = F.linear(x.transpose(1, 2).reshape(b, t, c * f), R._qw(P[pfx + "out.weight"], cfg), P[pfx + "out.bias"])
    return x, cur.long()


def swish_conv_module(P, pfx, x: Tensor, valid: Tensor, kernel: int, bn_training: bool, cfg=None):
    """ConformerConvolution(pointwise_activation='swish'): all of the depthwise / BatchNorm work on 2*d channels.
    bf16 storage points: pointwise_conv1 output, Swish * mask, depthwise output (batch statistics from its fp32 accumulators),
    BatchNorm + Swish output, the two pointwise weight images."""
    h = R._q(F.linear(x, R._qw(P[pfx + "pointwise_conv1.weight"], cfg).squeeze(-1), P[pfx + "pointwise_conv1.bias"]), cfg)  # [B,T,2d]
    h = h * torch.sigmoid(h)
    h = R._q(h * valid.unsqueeze(-1).to(h.dtype), cfg)
    c2 = h.shape[-1]
    pad = (kernel - 1) // 2
    c = F.conv1d(F.pad(h.transpose(1, 2), (pad, pad)), P[pfx + "depthwise_conv.weight"], P[pfx + "depthwise_conv.bias"],
                 groups=c2)
    cq = R._q(c, cfg)
    if bn_training:
        mean, var = c.mean(dim=(0, 2)), c.var(dim=(0, 2), unbiased=False)
    else:
        mean, var = P[pfx + "batch_norm.running_mean"], P[pfx + "batch_norm.running_var"]
    c = (cq - mean.view(1, c2, 1)) * torch.rsqrt(var.view(1, c2, 1) + 1e-5)
    c = c * P[pfx + "batch_norm.weight"].view(1, c2, 1) + P[pfx + "batch_norm.bias"].view(1, c2, 1)
    c = R._q(c * torch.sigmoid(c), cfg)
    return F.linear(c.transpose(1, 2), R._qw(P[pfx + "pointwise_conv2.weight"], cfg).squeeze(-1), P[pfx + "pointwise_conv2.bias"])


def _sb(P, pfx, x, cfg=None):
    return R._q(x * P[pfx + "scale"] + P[pfx + "bias"], cfg)  # ScaleBias writes the next GEMM's (bf16) operand


def squeezeformer_layer(P, pfx, cfg: SqueezeformerCfg, x, pos_emb, valid, bn_training):
    acfg = R.ConformerCfg(d_model=cfg.d_model, n_heads=cfg.n_heads, n_layers=1, dropout=0, dropout_att=cfg.dropout_att,
                          dropout_pre_encoder=0, emulate_bf16=cfg.emulate_bf16)
    x = R._ln(P, pfx + "norm_self_att.", x + R.rel_pos_attention(P, pfx + "self_attn.", acfg, _sb(P, pfx + "self_attn_scale.", x, cfg),
                                                               pos_emb, valid, False))
    x = R._ln(P, pfx + "norm_feed_forward1.", x + R.feed_forward(P, pfx + "feed_forward1.", acfg,
                                                                _sb(P, pfx + "feed_forward1_scale.", x, cfg), False))
    x = R._ln(P, pfx + "norm_conv.", x + swish_conv_module(P, pfx + "conv.", _sb(P, pfx + "conv_scale.", x, cfg), valid,
                                                          cfg.conv_kernel, bn_training, cfg))
    x = R._ln(P, pfx + "norm_feed_forward2.", x + R.feed_forward(P, pfx + "feed_forward2.", acfg,
                                                                _sb(P, pfx + "feed_forward2_scale.", x, cfg), False))
    return x


def time_reduction(P, pfx, x: Tensor, valid: Tensor, cfg=None):
    """x [B,T,d], valid [B,T] -> ([B, ceil(T/2), d], valid[:, ::2]); bf16 storage: the depthwise output, the pointwise weights"""
    h = (x * valid.unsqueeze(-1).to(x.dtype)).transpose(1, 2)
    d = h.shape[1]
    h = R._q(F.conv1d(h, P[pfx + "dw_conv.weight"], P[pfx + "dw_conv.bias"], stride=2, padding=3, groups=d), cfg)
    h = F.conv1d(h, R._qw(P[pfx + "pw_conv.weight"], cfg), P[pfx + "pw_conv.bias"]).transpose(1, 2)
    v2 = valid[:, ::2]
    return F.pad(h, (0, 0, 0, v2.shape[1] - h.shape[1])), v2


def encoder_forward(P: Dict[str, Tensor], cfg: SqueezeformerCfg, mel: Tensor, mel_len: Tensor, bn_training: bool = False):
    """-> (encoded [B, d, T'], lengths [B]); dropout = 0 (parity configuration)"""
    x, enc_len = dw_striding_forward(P, mel, mel_len, cfg=cfg)
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
            x, valid = time_reduction(P, "time_reduce_layer.", x, valid, cfg)
            pos_emb = R.rel_pos_table(x.shape[1], d).to(x.dtype)
        if cfg.time_reduce_idx is not None and i == rec_idx:
            x0, valid, pos_emb = cache
            x = torch.repeat_interleave(R._q(x, cfg), repeats=2, dim=1)[:, : x0.shape[1]]
            x = x0 + F.linear(x, R._qw(P["time_recovery_layer.weight"], cfg), P["time_recovery_layer.bias"])
        x = squeezeformer_layer(P, f"layers.{i}.", cfg, x, pos_emb, valid, bn_training)
    return x.transpose(1, 2), enc_len
