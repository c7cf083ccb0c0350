"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain PyTorch fp32) of the reference hot path.

This is the *oracle* the HIP path is checked against on the GPU box, where /root/reference does
not exist.  It is a functional restatement (parameters in a flat dict keyed exactly like the
reference `state_dict`, the on-disk ABI of SURVEY.md section 8b) of:

  * mel front-end ............ nemo/collections/asr/parts/preprocessing/features.py:59-93, 386-502
  * conv subsampling (x4) .... nemo/collections/asr/parts/submodules/subsampling.py:385-436, 576-586, 725-759
  * rel. positional encoding . nemo/collections/asr/parts/submodules/multi_head_attention.py:1015-1100
  * Conformer layer .......... nemo/collections/asr/parts/submodules/conformer_modules.py:160-233, 320-350, 382-387
  * rel-pos MHA .............. nemo/collections/asr/parts/submodules/multi_head_attention.py:124-146, 259-354
  * encoder orchestration .... nemo/collections/asr/modules/conformer_encoder.py:593-759, 794-848
  * decoder .................. nemo/collections/asr/modules/conv_asr.py:445-468
  * CTC loss ................. nemo/collections/asr/losses/ctc.py:45-82 (arithmetic = torch.nn.functional.ctc_loss)

Pinned: tests/test_oracle_pinning.py checks it against the reference's own files executed through
`oracle/ref_shim.py` (in the build container) and against the committed fixtures in tests/golden/
(everywhere).  CTC is additionally pinned by the warp-ctc known-answer vectors the reference keeps in
tests/collections/asr/k2/test_ctc.py (see oracle/ctc_ref.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
INF_VAL = 10000.0  # multi_head_attention.py:52
LOG_GUARD = 2.0 ** -24  # features.py:265
STD_EPS = 1e-5  # features.py:56 CONSTANT


@dataclass
class ConformerCfg:
    feat_in: int = 80
    d_model: int = 512
    n_heads: int = 8
    n_layers: int = 18
    ff_expansion: int = 4
    conv_kernel: int = 31
    vocab: int = 128  # blank id == vocab
    conv_channels: int = -1  # -1 => d_model
    dropout: float = 0.1
    dropout_pre_encoder: float = 0.1
    dropout_att: float = 0.1
    xscaling: bool = True
    # bf16 emulation: round (in fp32 arithmetic) at the points where the MI355X bf16 path keeps bf16 in HBM -- GEMM operand
    # images of the weights, LayerNorm outputs, every saved activation between kernels -- and round the gradient flowing back
    # through the same points.  Accumulation stays fp32 (as in the MFMA).  Lets a test separate "bf16 storage rounding" from
    # "kernel error": tests/test_baseline_configs_gpu.py
    emulate_bf16: bool = False
    # ---- encoder options of the sibling recipes (conf/fastconformer/cache_aware_streaming/*, long-form and InterCTC recipes);
    # restated here FIRST (oracle before kernel, pinned to the reference by tests/test_oracle_pinning.py and
    # tests/golden/ref_encoder_options.npz); the MI355X encoder still raises NotImplementedError for the non-default values.
    att_context_size: Tuple[int, int] = (-1, -1)   # [left, right] frames visible to a query; -1 = unlimited (conformer_encoder.py:794-823)
    att_context_style: str = "regular"             # "regular" | "chunked_limited"
    conv_norm_type: str = "batch_norm"             # "batch_norm" | "layer_norm" (conformer_modules.py:293-306, 335-340)
    conv_context_size: Optional[Tuple[int, int]] = None  # [left, right] padding of the depthwise conv; None = symmetric; (k-1, 0) = causal
    feat_out: int = -1                              # > 0 and != d_model: a Linear(d_model, feat_out) after the last layer (conformer_encoder.py:474-479, 738-739)
    stochastic_depth_drop_prob: float = 0.0         # layers dropped at random in training (conformer_encoder.py:696-707, arXiv 2102.03216)
    stochastic_depth_mode: str = "linear"
    stochastic_depth_start_layer: int = 1
    self_attention_model: str = "rel_pos"           # "rel_pos" | "rel_pos_local_attn" (Longformer-style sliding window with the
                                                    # relative-position term inside the window, multi_head_attention.py:357-586; the
                                                    # long-form recipes: conf/fastconformer/long_fastconformer/*.yaml, window [128, 128])
    causal_downsampling: bool = False               # CausalConv2D in the sub-sampling: pad (k - 1, stride - 1) on time AND frequency,
                                                    # no symmetric padding (causal_convs.py:24-72, subsampling.py:147-149, 222-224)

    @property
    def channels(self):
        return self.d_model if self.conv_channels == -1 else self.conv_channels

    @property
    def d_ff(self):
        return self.d_model * self.ff_expansion

    @property
    def d_k(self):
        return self.d_model // self.n_heads

    @staticmethod
    def small(**kw):
        return ConformerCfg(d_model=176, n_heads=4, n_layers=16, **kw)

    @staticmethod
    def large(**kw):
        return ConformerCfg(d_model=512, n_heads=8, n_layers=18, **kw)


# ------------------------------------------------------------------------------------------------
# mel front-end
# ------------------------------------------------------------------------------------------------
def hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f * 3.0 / 200.0
    log = 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * 27.0 / np.log(6.4)
    return np.where(f >= 1000.0, log, lin)


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((m - 15.0) * np.log(6.4) / 27.0), m * 200.0 / 3.0)


def mel_filterbank(sr=16000, n_fft=512, n_mels=80, fmin=0.0, fmax=None) -> np.ndarray:
    """Slaney-scale, area-normalised triangular filterbank == librosa.filters.mel(norm='slaney')
    (third-party librosa>=0.10.1; call site features.py:338-344).  Returns fp32 [n_mels, n_fft//2+1]."""
    fmax = sr / 2.0 if fmax is None else fmax
    edges = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    bins = np.arange(n_fft // 2 + 1, dtype=np.float64) * sr / n_fft
    fb = np.zeros((n_mels, bins.size))
    for i in range(n_mels):
        up = (bins - edges[i]) / (edges[i + 1] - edges[i])
        down = (edges[i + 2] - bins) / (edges[i + 2] - edges[i + 1])
        fb[i] = np.maximum(0.0, np.minimum(up, down)) * (2.0 / (edges[i + 2] - edges[i]))
    return fb.astype(np.float32)


def hann_window_sym(n: int) -> Tensor:
    """torch.hann_window(n, periodic=False) (features.py:326)."""
    k = torch.arange(n, dtype=torch.float64)
    return (0.5 - 0.5 * torch.cos(2.0 * math.pi * k / (n - 1))).float()


def mel_seq_len(audio_len: Tensor, n_fft=512, hop=160) -> Tensor:
    """features.py:413-417 with center padding n_fft//2 each side."""
    return torch.div(audio_len + 2 * (n_fft // 2) - n_fft, hop, rounding_mode="floor").long()


def log_mel_features(
    audio: Tensor,
    audio_len: Tensor,
    fb: Optional[Tensor] = None,
    window: Optional[Tensor] = None,
    n_fft: int = 512,
    hop: int = 160,
    win: int = 400,
    preemph: float = 0.97,
    n_mels: int = 80,
    normalize=True,
    noise: Optional[Tensor] = None,
    dither: float = 0.0,
    pad_to: int = 0,
    pad_value: float = 0.0,
) -> Tuple[Tensor, Tensor]:
    """audio [B,S] f32, audio_len [B] -> (features [B,n_mels,T] with T = 1 + S//hop, seq_len [B]).

    Explicit framing + rfft instead of torch.stft, so the framing semantics (centre zero-pad n_fft//2,
    window zero-padded *centred* to n_fft, one-sided, unnormalised) are restated, not inherited.
    `noise` (same shape as audio) is the dither sample (features.py:435-436) when dither > 0.
    `normalize`: True / "per_feature" (the Conformer recipes), "all_features" (one mean / std per utterance over all valid cells,
    features.py:94-102), False / None / any other string such as the streaming recipes' "NA" (no normalisation, :111-112);
    `pad_to` > 0 pads the frame axis to a multiple of it with `pad_value` (:490-501)."""
    B, S = audio.shape
    x = audio.to(torch.float32)
    if dither > 0 and noise is not None:
        x = x + dither * noise
    seq_len = mel_seq_len(audio_len, n_fft, hop)
    seq_len = torch.where(audio_len == 0, torch.zeros_like(seq_len), seq_len)
    # pre-emphasis then zero beyond the true length (features.py:439-442)
    y = torch.cat([x[:, :1], x[:, 1:] - preemph * x[:, :-1]], dim=1)
    t = torch.arange(S).unsqueeze(0)
    y = torch.where(t < audio_len.unsqueeze(1), y, torch.zeros_like(y))
    # framing
    T = 1 + S // hop
    pad = n_fft // 2
    yp = F.pad(y, (pad, pad))
    idx = (torch.arange(T) * hop).unsqueeze(1) + torch.arange(n_fft).unsqueeze(0)  # [T, n_fft]
    frames = yp[:, idx]  # [B, T, n_fft]
    w = hann_window_sym(win) if window is None else window.float()
    wpad = torch.zeros(n_fft)
    off = (n_fft - win) // 2
    wpad[off : off + win] = w
    spec = torch.fft.rfft(frames * wpad, n=n_fft, dim=-1)  # [B, T, 257]
    power = spec.real ** 2 + spec.imag ** 2  # sqrt(.)**2 of the reference, mag_power = 2
    fbt = torch.from_numpy(mel_filterbank(n_fft=n_fft, n_mels=n_mels)) if fb is None else fb.reshape(n_mels, -1).float()
    mel = torch.matmul(power, fbt.t()).transpose(1, 2)  # [B, n_mels, T]
    feat = torch.log(mel + LOG_GUARD)
    tmask = (torch.arange(T).unsqueeze(0) < seq_len.unsqueeze(1)).unsqueeze(1)  # [B,1,T]
    if normalize == "all_features":
        for b in range(B):
            v = feat[b, :, : int(seq_len[b])]
            feat[b] = (feat[b] - v.mean()) / (v.std() + STD_EPS)
    elif normalize is True or normalize == "per_feature":
        n = seq_len.to(torch.float32).view(B, 1)
        mean = torch.where(tmask, feat, torch.zeros_like(feat)).sum(2) / n
        var = (torch.where(tmask, feat - mean.unsqueeze(2), torch.zeros_like(feat)) ** 2).sum(2) / (n - 1.0)
        std = torch.sqrt(var)
        std = torch.where(torch.isnan(std), torch.zeros_like(std), std) + STD_EPS
        feat = (feat - mean.unsqueeze(2)) / std.unsqueeze(2)
    feat = torch.where(tmask, feat, torch.full_like(feat, pad_value))
    if pad_to > 0 and feat.shape[-1] % pad_to:
        feat = F.pad(feat, (0, pad_to - feat.shape[-1] % pad_to), value=pad_value)
    return feat, seq_len


# ------------------------------------------------------------------------------------------------
# encoder
# ------------------------------------------------------------------------------------------------
def conv_out_len(n: Tensor, repeat: int = 2) -> Tensor:
    """subsampling.py:576-586 with kernel 3, stride 2, padding 1+1."""
    n = n.to(torch.float32)
    for _ in range(repeat):
        n = torch.floor((n + 2.0 - 3.0) / 2.0 + 1.0)
    return n.to(torch.int64)


def rel_pos_table(T: int, d: int) -> Tensor:
    """pos_emb [2T-1, d]; row r <-> relative position T-1-r (multi_head_attention.py:1015-1035,1067-1100)."""
    pos = torch.arange(T - 1, -T, -1, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(INF_VAL) / d))
    pe = torch.zeros(2 * T - 1, d)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


class _RoundBF16(torch.autograd.Function):
    """y = bf16(x) in value, and the gradient is rounded the same way on the way back (the HIP path stores both the
    activation and its gradient as bf16 at these points)"""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _RoundBF16Fwd(torch.autograd.Function):
    """bf16 operand image of an fp32 master weight: rounded in the forward, gradient passed through in fp32"""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGradBF16(torch.autograd.Function):
    """identity in the forward (the value stays an fp32 accumulator), the GRADIENT is stored as bf16 (e.g. the LSTM gate
    pre-activations and the joint's logits: f32 forward tensors whose gradients are bf16 GEMM operands)"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def _qg(x: Tensor, cfg) -> Tensor:
    return _RoundGradBF16.apply(x) if getattr(cfg, "emulate_bf16", False) else x


def _q(x: Tensor, cfg) -> Tensor:
    return _RoundBF16.apply(x) if getattr(cfg, "emulate_bf16", False) else x


def _qw(w: Tensor, cfg) -> Tensor:
    return _RoundBF16Fwd.apply(w) if getattr(cfg, "emulate_bf16", False) else w


def _drop(x: Tensor, p: float, train: bool) -> Tensor:
    return F.dropout(x, p, training=True) if (train and p > 0) else x


def subsampling_forward(P: Dict[str, Tensor], cfg: ConformerCfg, mel: Tensor, mel_len: Tensor, pfx="pre_encode."):
    """mel [B, F, T] -> [B, T', d]; masks as MaskedConvSequential (subsampling.py:725-759)."""
    B, Fdim, T = mel.shape
    x = mel.transpose(1, 2).unsqueeze(1)  # [B,1,T,F]

    def tmask(n, L):
        return (torch.arange(L).unsqueeze(0) < n.unsqueeze(1)).to(x.dtype).view(B, 1, L, 1)

    causal = getattr(cfg, "causal_downsampling", False)

    def conv(t, w, b):  # Conv2d(k 3, s 2, p 1), or CausalConv2D: F.pad (2, 1, 2, 1) then no padding
        return F.conv2d(F.pad(t, (2, 1, 2, 1)), w, b, stride=2) if causal else F.conv2d(t, w, b, stride=2, padding=1)

    def out_len(n):  # calculate_conv_output_size (subsampling.py:720-722): paddings (1, 1) or (2, 1), kernel 3, stride 2
        return torch.div(n + 3 - 3, 2, rounding_mode="floor") + 1 if causal else conv_out_len(n, 1)

    l0 = mel_len.to(torch.int64)
    x = x * tmask(l0, T)
    x = conv(x, P[pfx + "conv.0.weight"], P[pfx + "conv.0.bias"])  # (conv1 reads fp32 weights)
    l1 = out_len(l0)
    x = _q(torch.relu(x * tmask(l1, x.shape[2])) * tmask(l1, x.shape[2]), cfg)
    x = conv(x, _qw(P[pfx + "conv.2.weight"], cfg), P[pfx + "conv.2.bias"])
    l2 = out_len(l1)
    x = _q(torch.relu(x * tmask(l2, x.shape[2])) * tmask(l2, x.shape[2]), cfg)
    b, c, t, f = x.shape
    x = x.transpose(1, 2).reshape(b, t, c * f)
    x = F.linear(x, _qw(P[pfx + "out.weight"], cfg), P[pfx + "out.bias"])
    return x, l2


def context_mask(cfg, T: int) -> Tensor:
    """[T, T] bool, True = query i may see key j (ConformerEncoder._create_masks, conformer_encoder.py:794-823):
    'regular': -left <= j - i <= right (each side only if >= 0); 'chunked_limited': keys of the query's own chunk (chunk size =
    right + 1) and of the left // chunk_size chunks before it; with right == -1 it degenerates to the left-limited regular mask."""
    left, right = cfg.att_context_size
    i = torch.arange(T).unsqueeze(1)
    j = torch.arange(T).unsqueeze(0)
    ok = torch.ones(T, T, dtype=torch.bool)
    if cfg.att_context_style == "regular":
        if left >= 0:
            ok &= (j - i) >= -left
        if right >= 0:
            ok &= (j - i) <= right
    elif cfg.att_context_style == "chunked_limited":
        if right == -1:
            if left >= 0:
                ok &= (j - i) >= -left
        else:
            chunk = right + 1
            left_chunks = left // chunk if left >= 0 else 10000
            dc = torch.div(i, chunk, rounding_mode="trunc") - torch.div(j, chunk, rounding_mode="trunc")
            ok &= (dc <= left_chunks) & (dc >= 0)
    else:
        raise ValueError(f"att_context_style={cfg.att_context_style}")
    return ok


def rel_pos_attention(P, pfx, cfg: ConformerCfg, x: Tensor, pos_emb: Tensor, valid: Tensor, train: bool):
    """x [B,T,d] (already layer-normed); valid [B,T] bool.  score[b,h,i,j] = ((q_i+u)k_j + (q_i+v)p_{T-1+j-i})/sqrt(dk)."""
    B, T, d = x.shape
    H, dk = cfg.n_heads, cfg.d_k
    q = _q(F.linear(x, _qw(P[pfx + "linear_q.weight"], cfg), P[pfx + "linear_q.bias"]), cfg).view(B, T, H, dk)
    k = _q(F.linear(x, _qw(P[pfx + "linear_k.weight"], cfg), P[pfx + "linear_k.bias"]), cfg).view(B, T, H, dk).transpose(1, 2)
    v = _q(F.linear(x, _qw(P[pfx + "linear_v.weight"], cfg), P[pfx + "linear_v.bias"]), cfg).view(B, T, H, dk).transpose(1, 2)
    p = _q(F.linear(_q(pos_emb, cfg), _qw(P[pfx + "linear_pos.weight"], cfg)), cfg).view(2 * T - 1, H, dk).transpose(0, 1)  # [H,2T-1,dk]
    qu = _q(q + P[pfx + "pos_bias_u"], cfg).transpose(1, 2)  # [B,H,T,dk]
    qv = _q(q + P[pfx + "pos_bias_v"], cfg).transpose(1, 2)
    ac = torch.matmul(qu, k.transpose(-2, -1))  # [B,H,T,T]
    bd_full = torch.matmul(qv, p.transpose(-2, -1).unsqueeze(0))  # [B,H,T,2T-1]
    # explicit index map instead of the pad/view trick: bd[i,j] = bd_full[i, T-1+j-i]
    ii = torch.arange(T).unsqueeze(1)
    jj = torch.arange(T).unsqueeze(0)
    bd = bd_full[:, :, ii, T - 1 + jj - ii]
    scores = (ac + bd) / math.sqrt(dk)
    masked = ~(valid.unsqueeze(1) & valid.unsqueeze(2) & context_mask(cfg, T).unsqueeze(0))  # [B,T,T] True = masked
    masked = masked.unsqueeze(1)
    scores = scores.masked_fill(masked, -INF_VAL)
    attn = torch.softmax(scores, dim=-1).masked_fill(masked, 0.0)
    attn = _q(_drop(attn, cfg.dropout_att, train), cfg)  # the probabilities are an MFMA operand
    ctx = _q(torch.matmul(attn, v).transpose(1, 2).reshape(B, T, d), cfg)
    return F.linear(ctx, _qw(P[pfx + "linear_out.weight"], cfg), P[pfx + "linear_out.bias"])


def rel_pos_local_attention(P, pfx, cfg: ConformerCfg, x: Tensor, valid: Tensor, train: bool):
    """RelPositionMultiHeadAttentionLongformer.forward without global tokens (multi_head_attention.py:419-586), as the dense banded
    attention its overlapping-chunk arithmetic computes: query i sees keys j with |j - i| <= w, inside the sequence and not padded;
    score = ((q_i + u) . k_j + (q_i + v) . p_{j-i+w}) / sqrt(d_k) with p = linear_pos(pe), pe = sinusoid of the relative positions
    w ... -w (LocalAttRelPositionalEncoding :1103-1148: index t <-> position w - t = i - j); softmax over the window; rows of padded
    queries are zeroed (:542) -- their output is the bias of linear_out.  The reference adds the positional term diagonal by diagonal
    (:466-471) in a way that is only consistent for left == right, which is what the recipes use; other windows are refused here."""
    B, T, d = x.shape
    H, dk = cfg.n_heads, cfg.d_k
    left, right = cfg.att_context_size
    if left != right or left <= 0:
        raise ValueError("rel_pos_local_attn: att_context_size = [w, w] with w > 0")
    w = left
    lin = lambda name, t: F.linear(t, _qw(P[pfx + name + ".weight"], cfg), P[pfx + name + ".bias"])
    q = lin("linear_q", x).view(B, T, H, dk).transpose(1, 2)
    k = lin("linear_k", x).view(B, T, H, dk).transpose(1, 2)
    v = lin("linear_v", x).view(B, T, H, dk).transpose(1, 2)
    pe = rel_pos_table(w + 1, d).to(x.dtype)                                                  # positions w ... -w
    p = F.linear(pe, _qw(P[pfx + "linear_pos.weight"], cfg)).view(2 * w + 1, H, dk).transpose(0, 1)   # [H, 2w+1, dk]
    qu = q + P[pfx + "pos_bias_u"].unsqueeze(1)
    qv = q + P[pfx + "pos_bias_v"].unsqueeze(1)
    ac = torch.matmul(qu, k.transpose(-2, -1))                                                 # [B,H,T,T]
    bd_band = torch.matmul(qv, p.transpose(-2, -1).unsqueeze(0))                               # [B,H,T,2w+1]: diagonal c <-> j - i = c - w
    ii = torch.arange(T).unsqueeze(1)
    jj = torch.arange(T).unsqueeze(0)
    off = jj - ii
    inside = off.abs() <= w
    bd = bd_band[:, :, ii, (off + w).clamp(0, 2 * w)]
    scores = (ac + bd) / math.sqrt(dk)
    visible = inside.view(1, 1, T, T) & valid.view(B, 1, 1, T)
    scores = scores.masked_fill(~visible, float("-inf"))
    attn = torch.softmax(scores, dim=-1)
    attn = torch.nan_to_num(attn, nan=0.0).masked_fill(~valid.view(B, 1, T, 1), 0.0)
    attn = _drop(attn, cfg.dropout_att, train)
    ctx = torch.matmul(attn, v).transpose(1, 2).reshape(B, T, d)
    return lin("linear_out", ctx)


def conv_module(P, pfx, cfg: ConformerCfg, x: Tensor, valid: Tensor, bn_training: bool,
                bn_stats_out: Optional[dict] = None):
    """x [B,T,d] -> [B,T,d] (conformer_modules.py:320-350).  BN statistics over all B*T positions."""
    d = cfg.d_model
    h = _q(F.linear(x, _qw(P[pfx + "pointwise_conv1.weight"], cfg).squeeze(-1), P[pfx + "pointwise_conv1.bias"]), cfg)  # [B,T,2d]
    g = h[..., :d] * torch.sigmoid(h[..., d:])
    g = _q(g * valid.unsqueeze(-1).to(g.dtype), cfg)
    pad = (cfg.conv_kernel - 1) // 2
    lpad, rpad = (pad, pad) if cfg.conv_context_size is None else cfg.conv_context_size  # CausalConv1D, causal_convs.py:89-150
    c = F.conv1d(F.pad(g.transpose(1, 2), (lpad, rpad)), P[pfx + "depthwise_conv.weight"],
                 P[pfx + "depthwise_conv.bias"], groups=d)  # [B,d,T]
    cq = _q(c, cfg)  # batch statistics come from the fp32 accumulators, the stored tensor is bf16
    if cfg.conv_norm_type == "layer_norm":  # nn.LayerNorm over the channels of every frame (conformer_modules.py:335-338)
        c = F.layer_norm(cq.transpose(1, 2), (d,), P[pfx + "batch_norm.weight"], P[pfx + "batch_norm.bias"], 1e-5)
        c = _q(c * torch.sigmoid(c), cfg)
        return F.linear(c, _qw(P[pfx + "pointwise_conv2.weight"], cfg).squeeze(-1), P[pfx + "pointwise_conv2.bias"])
    if cfg.conv_norm_type != "batch_norm":
        raise ValueError(f"conv_norm_type={cfg.conv_norm_type}")
    if bn_training:
        mean = c.mean(dim=(0, 2))
        var = c.var(dim=(0, 2), unbiased=False)
        if bn_stats_out is not None:
            n = c.shape[0] * c.shape[2]
            bn_stats_out[pfx] = (mean.detach(), (var * n / max(n - 1, 1)).detach())
    else:
        mean, var = P[pfx + "batch_norm.running_mean"], P[pfx + "batch_norm.running_var"]
    c = (cq - mean.view(1, d, 1)) * torch.rsqrt(var.view(1, d, 1) + 1e-5)
    c = c * P[pfx + "batch_norm.weight"].view(1, d, 1) + P[pfx + "batch_norm.bias"].view(1, d, 1)
    c = _q(c * torch.sigmoid(c), cfg)
    return F.linear(c.transpose(1, 2), _qw(P[pfx + "pointwise_conv2.weight"], cfg).squeeze(-1), P[pfx + "pointwise_conv2.bias"])


def feed_forward(P, pfx, cfg, x, train):
    h = F.linear(x, _qw(P[pfx + "linear1.weight"], cfg), P[pfx + "linear1.bias"])
    h = _q(_drop(h * torch.sigmoid(h), cfg.dropout, train), cfg)  # Swish on the fp32 accumulator, stored as bf16
    return F.linear(h, _qw(P[pfx + "linear2.weight"], cfg), P[pfx + "linear2.bias"])


def _ln(P, pfx, x):
    return F.layer_norm(x, (x.shape[-1],), P[pfx + "weight"], P[pfx + "bias"], 1e-5)


def conformer_layer(P, pfx, cfg: ConformerCfg, x, pos_emb, valid, train, bn_training, bn_stats_out=None):
    r = x + 0.5 * _drop(feed_forward(P, pfx + "feed_forward1.", cfg, _q(_ln(P, pfx + "norm_feed_forward1.", x), cfg), train), cfg.dropout, train)
    if cfg.self_attention_model == "rel_pos_local_attn":
        att = rel_pos_local_attention(P, pfx + "self_attn.", cfg, _q(_ln(P, pfx + "norm_self_att.", r), cfg), valid, train)
    else:
        att = rel_pos_attention(P, pfx + "self_attn.", cfg, _q(_ln(P, pfx + "norm_self_att.", r), cfg), pos_emb, valid, train)
    r = r + _drop(att, cfg.dropout, train)
    r = r + _drop(conv_module(P, pfx + "conv.", cfg, _q(_ln(P, pfx + "norm_conv.", r), cfg), valid, bn_training, bn_stats_out), cfg.dropout, train)
    r = r + 0.5 * _drop(feed_forward(P, pfx + "feed_forward2.", cfg, _q(_ln(P, pfx + "norm_feed_forward2.", r), cfg), train), cfg.dropout, train)
    return _ln(P, pfx + "norm_out.", r)


def layer_drop_probs(cfg: ConformerCfg):
    """compute_stochastic_depth_drop_probs (parts/utils/regularization_utils.py:18-64): the first `start_layer` layers are never
    dropped; 'linear': l / L * p for the l-th of the remaining L layers, 'uniform': p for each of them."""
    p, start, n = cfg.stochastic_depth_drop_prob, cfg.stochastic_depth_start_layer, cfg.n_layers
    if not (0 <= p < 1.0):
        raise ValueError("stochastic_depth_drop_prob has to be in [0, 1).")
    if not (1 <= start <= n):
        raise ValueError("stochastic_depth_start_layer has to be in [1, num layers].")
    probs = [0.0] * start
    L = n - start
    if L > 0:
        if cfg.stochastic_depth_mode == "linear":
            probs += [l / L * p for l in range(1, L + 1)]
        elif cfg.stochastic_depth_mode == "uniform":
            probs += [p] * L
        else:
            raise ValueError(f'stochastic_depth_mode has to be one of ["linear", "uniform"]. Current value: {cfg.stochastic_depth_mode}')
    return probs


def encoder_forward(P, cfg: ConformerCfg, mel, mel_len, train=False, bn_training=None, pfx="", bn_stats_out=None,
                    n_layers: Optional[int] = None, capture: Optional[dict] = None, bypass_pre_encode: bool = False,
                    dropped: Optional[list] = None):
    """-> (encoded [B, d, T'], enc_len [B]).  `P` keys = reference encoder state_dict keys (+ optional prefix).
    `capture` = {layer index: None}: filled with that layer's output [B, d, T'] (0-based, after norm_out) -- what the reference
    registers as `interctc/layer_output_<l>` (conformer_encoder.py:724-736)."""
    bn_training = train if bn_training is None else bn_training
    if bypass_pre_encode:  # `mel` is already [B, T', d_model] (conformer_encoder.py:602-611, 630); a wrong shape is a ValueError (:569-578)
        if mel.shape[-1] != cfg.d_model:
            raise ValueError(f"If bypass_pre_encode is True, audio_signal should have shape (batch, n_frame, {cfg.d_model})")
        x, enc_len = mel, mel_len.to(torch.int64)
    else:
        x, enc_len = subsampling_forward(P, cfg, mel, mel_len, pfx + "pre_encode.")
    B, T, d = x.shape
    if cfg.xscaling:
        x = x * math.sqrt(d)
    x = _drop(x, cfg.dropout_pre_encoder, train)
    pos_emb = rel_pos_table(T, d).to(x.dtype)  # (float64 when the oracle is run in double to bound fp32 noise)
    valid = torch.arange(T).unsqueeze(0) < enc_len.unsqueeze(1)
    probs = layer_drop_probs(cfg) if cfg.stochastic_depth_drop_prob > 0.0 else None
    out_proj = (lambda t: F.linear(t, P[pfx + "out_proj.weight"], P[pfx + "out_proj.bias"])) \
        if (cfg.feat_out > 0 and cfg.feat_out != d) else (lambda t: t)
    for i in range(cfg.n_layers if n_layers is None else n_layers):
        x_in = x
        x = conformer_layer(P, f"{pfx}layers.{i}.", cfg, x, pos_emb, valid, train, bn_training, bn_stats_out)
        if train and probs is not None and probs[i] > 0.0:
            # one torch.rand(1) per droppable layer and forward, from the global generator, like the reference (:698); a dropped
            # layer still runs (x * 0 + input: every weight gets a gradient, no rank diverges), a kept one is rescaled by 1 / (1 - p)
            drop = bool(torch.rand(1) < probs[i])
            if dropped is not None:
                dropped.append(drop)
            x = x * 0.0 + x_in if drop else (x - x_in) / (1.0 - probs[i]) + x_in
        if capture is not None and i in capture:
            capture[i] = out_proj(x).transpose(1, 2)
    return out_proj(x).transpose(1, 2), enc_len


def decoder_forward(P, enc, pfx="decoder_layers.0.", cfg=None):
    """enc [B,d,T'] -> log-probs [B,T',V+1] (conv_asr.py:445-468)."""
    logits = F.linear(_q(enc.transpose(1, 2), cfg), _qw(P[pfx + "weight"], cfg).squeeze(-1), P[pfx + "bias"])
    return torch.log_softmax(logits, dim=-1)


def ctc_loss_mean_batch(logp, targets, in_len, tgt_len, blank):
    """losses/ctc.py:45-82 with reduction='mean_batch', zero_infinity=True."""
    per_utt = F.ctc_loss(logp.transpose(0, 1), targets.long(), in_len.long(), tgt_len.long(), blank=blank,
                         reduction="none", zero_infinity=True)
    return per_utt.mean(), per_utt


def model_forward(P, cfg: ConformerCfg, audio, audio_len, tokens, token_len, train=False, bn_training=None,
                  noise=None, dither=0.0, bn_stats_out=None, interctc: Optional[Tuple[list, list]] = None):
    """Full reference forward (ctc_models.py:495-546 + training_step loss :549-585), SpecAugment off.
    `P`: 'preprocessor.featurizer.fb/window' optional, 'encoder.*', 'decoder.decoder_layers.0.*'.
    `interctc` = (apply_at_layers, loss_weights): intermediate CTC losses through the SAME decoder on the captured layer outputs,
    loss = (1 - sum w) * final + sum_l w_l * inter_l (parts/mixins/interctc_mixin.py:46-58, 214-270; ctc_models.py:577-585)."""
    fb = P.get("preprocessor.featurizer.fb")
    window = P.get("preprocessor.featurizer.window")
    with torch.no_grad():
        mel, mel_len = log_mel_features(audio, audio_len, fb=fb, window=window, n_mels=cfg.feat_in,
                                        noise=noise, dither=dither)
    capture = {int(l): None for l in interctc[0]} if interctc else None
    enc, enc_len = encoder_forward(P, cfg, mel, mel_len, train=train, bn_training=bn_training, pfx="encoder.",
                                   bn_stats_out=bn_stats_out, capture=capture)
    logp = decoder_forward(P, enc, "decoder.decoder_layers.0.", cfg)
    loss, per_utt = ctc_loss_mean_batch(logp, tokens, enc_len, token_len, cfg.vocab)
    out = dict(per_utt=per_utt, logp=logp, enc=enc, enc_len=enc_len, mel=mel, mel_len=mel_len)
    if interctc:
        layers, weights = interctc
        if len(layers) != len(weights):
            raise ValueError("Length of interctc.apply_at_layers has to match interctc.loss_weights")
        out["final_loss"] = loss
        loss = loss * (1.0 - sum(weights))
        for l, w in zip(layers, weights):
            lp = decoder_forward(P, capture[int(l)], "decoder.decoder_layers.0.", cfg)
            inter, _ = ctc_loss_mean_batch(lp, tokens, enc_len, token_len, cfg.vocab)
            out[f"inter_ctc_loss_l{int(l)}"] = inter
            loss = loss + inter * w
    out["loss"] = loss
    return out


# ------------------------------------------------------------------------------------------------
# parameter init (shapes = the on-disk ABI; values: torch defaults, deterministic given the generator)
# ------------------------------------------------------------------------------------------------
def init_params(cfg: ConformerCfg, seed: int = 0, dtype=torch.float32, nonzero_pos_bias: bool = True) -> Dict[str, Tensor]:
    g = torch.Generator().manual_seed(seed)
    P: Dict[str, Tensor] = {}

    def uni(shape, bound):
        return (torch.rand(shape, generator=g, dtype=dtype) * 2 - 1) * bound

    def linear(name, out_f, in_f, bias=True, extra=()):
        bound = 1.0 / math.sqrt(in_f * int(np.prod(extra)) if extra else in_f)
        P[name + ".weight"] = uni((out_f, in_f) + tuple(extra), bound)
        if bias:
            P[name + ".bias"] = uni((out_f,), bound)

    d, C, Fi = cfg.d_model, cfg.channels, cfg.feat_in
    fo = Fi
    for _ in range(2):
        fo = (fo + 2 - 3) // 2 + 1
    e = "encoder."
    P[e + "pre_encode.conv.0.weight"] = uni((C, 1, 3, 3), 1 / 3.0)
    P[e + "pre_encode.conv.0.bias"] = uni((C,), 1 / 3.0)
    P[e + "pre_encode.conv.2.weight"] = uni((C, C, 3, 3), 1 / math.sqrt(9 * C))
    P[e + "pre_encode.conv.2.bias"] = uni((C,), 1 / math.sqrt(9 * C))
    linear(e + "pre_encode.out", d, C * fo)
    for i in range(cfg.n_layers):
        l = f"{e}layers.{i}."
        for n in ("norm_feed_forward1", "norm_self_att", "norm_conv", "norm_feed_forward2", "norm_out"):
            P[l + n + ".weight"] = 1.0 + 0.1 * uni((d,), 1.0)
            P[l + n + ".bias"] = 0.1 * uni((d,), 1.0)
        for ff in ("feed_forward1", "feed_forward2"):
            linear(l + ff + ".linear1", cfg.d_ff, d)
            linear(l + ff + ".linear2", d, cfg.d_ff)
        for n in ("linear_q", "linear_k", "linear_v", "linear_out"):
            linear(l + "self_attn." + n, d, d)
        linear(l + "self_attn.linear_pos", d, d, bias=False)
        pb = 0.1 if nonzero_pos_bias else 0.0  # reference init is zeros (multi_head_attention.py:248-254)
        P[l + "self_attn.pos_bias_u"] = uni((cfg.n_heads, cfg.d_k), 1.0) * pb
        P[l + "self_attn.pos_bias_v"] = uni((cfg.n_heads, cfg.d_k), 1.0) * pb
        linear(l + "conv.pointwise_conv1", 2 * d, d, extra=(1,))
        P[l + "conv.depthwise_conv.weight"] = uni((d, 1, cfg.conv_kernel), 1 / math.sqrt(cfg.conv_kernel))
        P[l + "conv.depthwise_conv.bias"] = uni((d,), 1 / math.sqrt(cfg.conv_kernel))
        P[l + "conv.batch_norm.weight"] = 1.0 + 0.1 * uni((d,), 1.0)
        P[l + "conv.batch_norm.bias"] = 0.1 * uni((d,), 1.0)
        P[l + "conv.batch_norm.running_mean"] = torch.zeros(d, dtype=dtype)
        P[l + "conv.batch_norm.running_var"] = torch.ones(d, dtype=dtype)
        P[l + "conv.batch_norm.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
        linear(l + "conv.pointwise_conv2", d, d, extra=(1,))
    V1 = cfg.vocab + 1
    bound = math.sqrt(6.0 / (d + V1))  # xavier_uniform (conv_asr.py:424,448)
    P["decoder.decoder_layers.0.weight"] = uni((V1, d, 1), bound)
    P["decoder.decoder_layers.0.bias"] = uni((V1,), 1 / math.sqrt(d))
    return P


def trainable_keys(P):
    return [k for k in P if not (k.endswith("running_mean") or k.endswith("running_var")
                                 or k.endswith("num_batches_tracked") or k.startswith("preprocessor."))]


def synthetic_batch(B: int, secs: float, vocab: int = 128, seed: int = 1234, lengths: Optional[Tensor] = None):
    """SURVEY.md section 8(d) synthetic inputs: audio 0.1*randn, tokens randint(0, vocab), U = 3*secs."""
    g = torch.Generator().manual_seed(seed)
    S = int(round(16000 * secs))
    audio = 0.1 * torch.randn(B, S, generator=g)
    audio_len = torch.full((B,), S, dtype=torch.int64) if lengths is None else lengths.to(torch.int64)
    U = max(1, int(3 * secs))
    tokens = torch.randint(0, vocab, (B, U), generator=g)
    token_len = torch.full((B,), U, dtype=torch.int64)
    return audio, audio_len, tokens, token_len


def grad_digest(name: str, g) -> np.ndarray:
    """compact fingerprint of one gradient tensor for fixtures of configurations whose gradients are too large to commit
    (Small: 13 M values): float64 (L2 norm, max |.|, projection on a fixed pseudo-random +-1 vector seeded by the name)"""
    import zlib
    g = np.asarray(g, dtype=np.float64).ravel()
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    sign = rs.randint(0, 2, size=g.size).astype(np.float64) * 2.0 - 1.0
    return np.array([np.sqrt((g * g).sum()), np.abs(g).max() if g.size else 0.0, (g * sign).sum()])


def grad_projections(name: str, g: Tensor, K: int = 16) -> Tensor:
    """K projections of a gradient tensor on fixed pseudo-random +-1 vectors (float64 [K]), for fixtures of runs whose gradients
    are too large to commit (Large: 121.5 M values): for independent sign vectors s_k, mean_k ((a - b) . s_k)^2 estimates
    ||a - b||^2 (relative spread sqrt(2 / K)), so the distance of a tensor to the committed run is recoverable from K numbers.
    The signs come from an integer hash of (element index, k, crc32(name)) evaluated with torch int64 arithmetic (wrapping
    multiplies, masked shifts): bit-identical on CPU and GPU."""
    import zlib
    g = g.detach().reshape(-1).to(torch.float64)
    n = g.numel()
    dev = g.device
    seed = zlib.crc32(name.encode()) & 0x7FFFFFFF
    idx = torch.arange(n, dtype=torch.int64, device=dev)
    out = torch.empty(K, dtype=torch.float64, device=dev)
    M1, M2 = -7046029254386353131, -4658895280553007687   # 0x9E3779B97F4A7C15, 0xBF58476D1CE4E5B9 as int64
    mask35, mask31 = (1 << 35) - 1, (1 << 33) - 1
    for k in range(K):
        h = (idx + (seed + 1000003 * k)) * M1
        h = h ^ ((h >> 29) & mask35)
        h = h * M2
        h = h ^ ((h >> 31) & mask31)
        sign = ((h >> 40) & 1).to(torch.float64) * 2.0 - 1.0
        out[k] = (g * sign).sum()
    return out
