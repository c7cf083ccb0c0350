"""TEST INFRASTRUCTURE ONLY -- the Conformer layers on the VALID frames of a ragged batch only ("packed token dimension").

SURVEY.md section 8 row f1 asks for length-aware kernels that skip padded frames; the `[B*T', d]` chain of the MI355X encoder
still computes every padded row (7 % of the step with semi-sorted batches, 42 % with unshaped ones).  This
file is the oracle for the packed form, written BEFORE the kernels: it runs the layers of `oracle/conformer_ref.py` on
`sum_b L_b` rows instead of `B * T'_max` and must reproduce the padded computation -- i.e. the reference's
(conformer_encoder.py:593-759, conformer_modules.py:160-350) -- on every valid frame, in the loss and in every gradient.

What makes that non-trivial is BatchNorm (conformer_modules.py:297, 339; SURVEY Appendix A: the batch statistics run over all
B * T' positions INCLUDING padded frames).  A padded frame is not "nothing":
  * the depthwise convolution (k taps, zero-masked input beyond the utterance: `masked_fill`, conformer_modules.py:330-331) still
    produces `bias + sum of the taps that reach valid frames` on the first (k-1)/2 frames after an utterance's end -- the HALO --
    and exactly `bias` on every padded frame beyond it;
  * so the statistics are: the computed sums over (valid + halo) frames of every utterance, plus `n_rest * bias` and
    `n_rest * bias^2` for the `n_rest = sum_b (T'_max - L_b - halo_b)` frames that are pure bias;
  * and their gradient flows back through the halo frames into the last valid frames of the utterance and into the depthwise bias
    and taps -- autograd does it here; a packed kernel has to add `-(k1 + xhat * k2) * gamma * rstd` on the halo rows and the
    closed-form bias term (the same k1 / k2 as mi355x_bn_swish_bwd_apply).
Everything else is row-wise (LayerNorm, feed-forward, projections, GLU, pointwise convs, residuals) or per utterance with masked
keys (attention: the relative-position table of a shorter utterance is the centre slice of the longer one's), so padded rows never
reach a valid row except through those statistics -- which this file makes explicit and tests/test_oracle_pinning.py checks.

Only tests/ may import this module.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

from . import conformer_ref as R

Tensor = torch.Tensor


def conv_module_packed(P: Dict[str, Tensor], pfx: str, cfg: R.ConformerCfg, xs: List[Tensor], t_max: int, bn_training: bool):
    """xs[b]: [L_b, d] (layer-normed valid rows) -> list of [L_b, d].  BatchNorm statistics as over the padded [B, d, t_max] tensor."""
    d, k = cfg.d_model, cfg.conv_kernel
    pad = (k - 1) // 2
    w, bias = P[pfx + "depthwise_conv.weight"], P[pfx + "depthwise_conv.bias"]
    ext, n_rest = [], 0
    for x in xs:
        L = x.shape[0]
        h = F.linear(x, P[pfx + "pointwise_conv1.weight"].squeeze(-1), P[pfx + "pointwise_conv1.bias"])
        g = h[:, :d] * torch.sigmoid(h[:, d:])                       # [L, d]; frames >= L are zero by the reference's mask
        halo = min(pad, t_max - L)                                    # frames after the end whose taps still reach valid frames
        c = F.conv1d(F.pad(g.t().unsqueeze(0), (pad, pad + halo)), w, bias, groups=d)[0]  # [d, L + halo]
        ext.append((c, L, halo))
        n_rest += t_max - L - halo
    if bn_training:
        n = len(xs) * t_max
        mean = (sum(c.sum(dim=1) for c, _, _ in ext) + n_rest * bias) / n
        # (two-pass variance in this fp32 restatement; the kernels carry sum and sum of squares in f64 and may add n_rest * bias^2)
        var = (sum(((c - mean.unsqueeze(1)) ** 2).sum(dim=1) for c, _, _ in ext) + n_rest * (bias - mean) ** 2) / n
    else:
        mean, var = P[pfx + "batch_norm.running_mean"], P[pfx + "batch_norm.running_var"]
    out = []
    for c, L, _ in ext:
        y = (c[:, :L] - mean.unsqueeze(1)) * torch.rsqrt(var.unsqueeze(1) + 1e-5)
        y = y * P[pfx + "batch_norm.weight"].unsqueeze(1) + P[pfx + "batch_norm.bias"].unsqueeze(1)
        y = y * torch.sigmoid(y)
        out.append(F.linear(y.t(), P[pfx + "pointwise_conv2.weight"].squeeze(-1), P[pfx + "pointwise_conv2.bias"]))
    return out


def encoder_layers_packed(P: Dict[str, Tensor], cfg: R.ConformerCfg, x: Tensor, enc_len: Tensor, bn_training: bool, pfx: str = ""):
    """x: [B, T', d] after sub-sampling / scaling (padded), enc_len [B] -> [B, T', d] with the VALID frames computed from valid
    frames only and zeros elsewhere (the padded rows of the reference hold values nothing downstream reads)."""
    if cfg.att_context_size != (-1, -1) or cfg.conv_norm_type != "batch_norm" or cfg.conv_context_size is not None:
        raise NotImplementedError("packed oracle: default encoder options only")
    B, T, d = x.shape
    lens = [int(n) for n in enc_len]
    xs = [x[b, :lens[b]] for b in range(B)]
    for i in range(cfg.n_layers):
        p = f"{pfx}layers.{i}."
        ln = lambda name, t: R._ln(P, p + name, t)
        xs = [t + 0.5 * R.feed_forward(P, p + "feed_forward1.", cfg, ln("norm_feed_forward1.", t), False) for t in xs]
        att = []
        for t in xs:
            L = t.shape[0]
            pos = R.rel_pos_table(L, d).to(t.dtype)  # = the centre 2L-1 rows of the T'-frame table
            a = R.rel_pos_attention(P, p + "self_attn.", cfg, ln("norm_self_att.", t).unsqueeze(0), pos,
                                    torch.ones(1, L, dtype=torch.bool), False)[0]
            att.append(t + a)
        xs = att
        cv = conv_module_packed(P, p + "conv.", cfg, [ln("norm_conv.", t) for t in xs], T, bn_training)
        xs = [t + c for t, c in zip(xs, cv)]
        xs = [t + 0.5 * R.feed_forward(P, p + "feed_forward2.", cfg, ln("norm_feed_forward2.", t), False) for t in xs]
        xs = [ln("norm_out.", t) for t in xs]
    out = x.new_zeros(B, T, d)
    rows = [F.pad(t, (0, 0, 0, T - t.shape[0])) for t in xs]
    return out + torch.stack(rows)


def encoder_forward_packed(P, cfg: R.ConformerCfg, mel, mel_len, bn_training=True, pfx=""):
    """the padded sub-sampling of the reference path, then the layers on valid frames only -> (encoded [B, d, T'], enc_len)"""
    x, enc_len = R.subsampling_forward(P, cfg, mel, mel_len, pfx + "pre_encode.")
    if cfg.xscaling:
        x = x * (x.shape[-1] ** 0.5)
    y = encoder_layers_packed(P, cfg, x, enc_len, bn_training, pfx)
    return y.transpose(1, 2), enc_len
