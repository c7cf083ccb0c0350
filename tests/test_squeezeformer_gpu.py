"""SURVEY.md section 8f row 2 / BASELINE.json configs[4]: the Squeezeformer encoder on the HIP engine against (1) the fixture made
by the reference's own SqueezeformerEncoder class and (2) the pinned oracle (oracle/squeezeformer_ref.py) under autograd at
geometries the fixture does not reach: d_model and d_k that are NOT multiples of 8 (Squeezeformer-Medium is d = 324, 4 heads,
d_k = 81), where the engine pads the heads inside the packed weight images and pitches the activations."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")

from oracle import squeezeformer_ref as SQ  # noqa: E402  (test infrastructure: the checker)


def _rel(a, b):
    return (a - b).norm().item() / max(b.norm().item(), 1e-12)


def test_squeezeformer_encoder_matches_reference_fixture(golden_dir):
    """tests/golden/ref_squeezeformer_tiny.npz: 'dw_striding' x4, adaptive scale/bias, Swish conv module on 2d channels with
    batch-statistics BatchNorm, time reduction at layer 1 / recovery at layer 3, ragged lengths: output, lengths and the
    gradient of a fixed linear functional w.r.t. EVERY parameter."""
    from nemo_amd.modules import SqueezeformerEncoder
    from test_model_gpu import _encoder_vs_fixture
    z = np.load(os.path.join(golden_dir, "ref_squeezeformer_tiny.npz"))
    enc = SqueezeformerEncoder(feat_in=40, n_layers=4, d_model=32, subsampling="dw_striding", subsampling_factor=4,
                               subsampling_conv_channels=-1, ff_expansion_factor=4, n_heads=4, conv_kernel_size=9, dropout=0.0,
                               dropout_emb=0.0, dropout_att=0.0, adaptive_scale=True, time_reduce_idx=1, time_recovery_idx=3)
    _encoder_vs_fixture(z, enc)


def _oracle_run(enc, cfg, x, length, w):
    P = {k: v.detach().double().cpu().requires_grad_(v.dtype.is_floating_point and k in dict(enc.named_parameters()))
         for k, v in enc.state_dict().items()}
    y, yl = SQ.encoder_forward(P, cfg, x.double().cpu(), length.cpu(), bn_training=True)
    valid = (torch.arange(y.shape[2]).unsqueeze(0) < yl.unsqueeze(1)).unsqueeze(1)
    (y * w.double() * valid).sum().backward()
    return y.detach(), yl, {k: v.grad for k, v in P.items() if v.requires_grad and v.grad is not None}, valid


@pytest.mark.parametrize("d_model,n_heads,T,extra", [(36, 4, 83, {}), (20, 2, 64, {}),
                                                     (24, 3, 70, dict(adaptive_scale=False, time_reduce_idx=None, time_recovery_idx=None)),
                                                     (24, 3, 61, dict(time_reduce_idx=0, time_recovery_idx=None))])
def test_squeezeformer_odd_geometry_fp32_matches_oracle(d_model, n_heads, T, extra):
    """d_k = 9 / 10 (padded to 16 inside the weight images), d_model % 8 != 0 (activation pitch 40 / 24), odd and even frame
    counts through the time reduction: forward and every parameter gradient against the float64 oracle."""
    from nemo_amd.modules import SqueezeformerEncoder
    torch.manual_seed(5)
    kw = dict(feat_in=24, n_layers=3, d_model=d_model, subsampling="dw_striding", subsampling_factor=4, n_heads=n_heads,
              conv_kernel_size=5, dropout=0.0, dropout_emb=0.0, dropout_att=0.0, time_reduce_idx=1, time_recovery_idx=2)
    kw.update(extra)  # (also: fixed scale / bias buffers, no temporal U-Net, reduction at layer 0 with recovery at the last layer)
    enc = SqueezeformerEncoder(compute_dtype=torch.float32, **kw)
    with torch.no_grad():
        for n, p in enc.named_parameters():
            if n.endswith("_scale.scale"):
                p.add_(0.2 * torch.randn_like(p))
            elif n.endswith("_scale.bias") or "pos_bias" in n:
                p.add_(0.1 * torch.randn_like(p))
    cfg = SQ.SqueezeformerCfg(feat_in=24, d_model=d_model, n_heads=n_heads, n_layers=3, conv_kernel=5,
                              time_reduce_idx=kw["time_reduce_idx"], time_recovery_idx=kw["time_recovery_idx"])
    x = torch.randn(3, 24, T)
    length = torch.tensor([T, T - 17, T // 2])
    T2 = ((T - 1) // 2 + 1 - 1) // 2 + 1
    w = torch.randn(3, d_model, T2)
    yr, ylr, gr, valid = _oracle_run(enc, cfg, x, length, w)
    enc = enc.to(dev).train()
    enc.flat_parameters().zero_grad()
    y, yl = enc(audio_signal=x.to(dev), length=length.to(dev))
    assert yl.tolist() == ylr.tolist()
    assert _rel((y.detach().cpu().double() * valid), yr * valid) < 2e-5
    (y * (w * valid).to(dev)).sum().backward()
    torch.cuda.synchronize()
    bad = {}
    for n, p in enc.named_parameters():
        e = _rel(p.grad.detach().cpu().double(), gr[n])
        if e > 1e-3 and gr[n].norm() > 1e-6:
            bad[n] = e
    assert not bad, bad


def test_squeezeformer_medium_geometry_bf16_tracks_fp32():
    """d_model = 324, 4 heads (d_k = 81 -> 88 lanes, activation pitch 328), 648-channel depthwise stage, time reduction and
    recovery, bf16 GEMMs (sub-sampling stack in fp32: 324 conv channels): output and flat gradient against the fp32 run of
    the same weights; then the recipe's dropout values (stochastic, finite)."""
    from nemo_amd.modules import SqueezeformerEncoder
    kw = dict(feat_in=80, n_layers=3, d_model=324, subsampling="dw_striding", subsampling_factor=4, n_heads=4,
              conv_kernel_size=31, dropout=0.0, dropout_emb=0.0, dropout_att=0.0, time_reduce_idx=1, time_recovery_idx=2)
    torch.manual_seed(6)
    e32 = SqueezeformerEncoder(compute_dtype=torch.float32, **kw)
    e16 = SqueezeformerEncoder(compute_dtype=torch.bfloat16, **kw)
    e16.load_state_dict(e32.state_dict())
    e32, e16 = e32.to(dev).train(), e16.to(dev).train()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, 80, 402, generator=g).to(dev)
    length = torch.tensor([402, 333, 150]).to(dev)
    outs = []
    for e in (e32, e16):
        e.flat_parameters().zero_grad()
        y, yl = e(audio_signal=x, length=length)
        assert y.shape == (3, 324, 101) and yl.tolist() == [101, 84, 38]
        (y.float() ** 2).mean().backward()
        torch.cuda.synchronize()
        outs.append((y.detach().float(), e.flat_parameters().grad.detach().clone()))
        assert torch.isfinite(outs[-1][1]).all()
    (y32, g32), (y16, g16) = outs
    assert _rel(y16, y32) < 3e-2
    assert torch.dot(g16, g32) / (g16.norm() * g32.norm()) > 0.99
    fp32_, fp16_ = e32.flat_parameters(), e16.flat_parameters()
    bad = {}
    for n in fp32_.order:  # per tensor, not only the flat cosine: a wrong head / pad lane would hide in the aggregate
        o, k = fp32_.offsets[n]
        a, b = g16[o:o + k], g32[o:o + k]
        if b.norm() > 1e-4 * g32.norm() and _rel(a, b) > 8e-2:
            bad[n] = _rel(a, b)
    assert not bad, bad
    kw.update(dropout=0.1, dropout_att=0.1)
    ed = SqueezeformerEncoder(compute_dtype=torch.bfloat16, **kw)
    ed.load_state_dict(e32.state_dict())
    ed = ed.to(dev).train()
    ya, _ = ed(audio_signal=x, length=length)
    yb, _ = ed(audio_signal=x, length=length)
    (ya.float() ** 2).mean().backward()
    torch.cuda.synchronize()
    assert torch.isfinite(ya).all() and not torch.equal(ya, yb)
    assert torch.isfinite(ed.flat_parameters().grad).all()


def test_squeezeformer_ctc_model_trains():
    """EncDecCTCModel built from the Squeezeformer recipe's model section (encoder _target_ resolved through the alias table,
    NoamHoldAnnealing schedule): a few fused optimizer steps lower the loss"""
    from nemo_amd.models import EncDecCTCModel, squeezeformer_ctc_config
    from oracle import conformer_ref as R
    cfg = squeezeformer_ctc_config("xs", vocab_size=32, n_layers=3, d_model=144, time_reduce_idx=1, dropout=0.0, dropout_att=0.0,
                                   compute_dtype=torch.bfloat16)
    cfg["preprocessor"]["dither"] = 0.0
    cfg["optim"]["lr"] = 1e-3
    cfg["optim"]["sched"]["warmup_steps"] = 2
    torch.manual_seed(1)
    m = EncDecCTCModel(cfg).to(dev).train()
    m.decoder.compute_dtype = torch.bfloat16
    m.setup_optimization(cfg["optim"])
    audio, alen, tok, tl = R.synthetic_batch(4, 2.0, vocab=32, seed=3)
    batch = [audio.to(dev), alen.to(dev), tok.to(dev), tl.to(dev)]
    losses = [m.fit_step(batch)["loss"].item() for _ in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < 0.9 * losses[0], losses
