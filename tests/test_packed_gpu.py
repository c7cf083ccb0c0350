"""-m gpu: the PACKED token chain ("length-aware kernels skipping padded frames").

With `encoder.packed_rows` on, the layers' row-wise chain (LayerNorm, feed-forward / projection / pointwise-conv GEMMs, residuals,
weight gradients) and the fused attention run on the sum_b L_b valid frames of a ragged batch; only the depthwise-conv + BatchNorm
core keeps the padded grid (the reference's batch statistics run over padded frames: conformer_modules.py:297,330-331).  Three-way
check on the four ragged batches of tests/test_oracle_pinning.py::test_packed_token_restatement_...: the packed HIP path against the
padded HIP path and against oracle/packed_ref.py (itself pinned to the padded restatement, which is pinned to the reference) --
loss and EVERY gradient within 1e-3 in fp32; then the bf16 production kernels (fused attention with row offsets) packed vs padded."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import conformer_ref as R

dev = "cuda"
LENGTHS = [[16000, 12000, 8123], [16000, 16000, 16000], [16000, 2400, 9000], [15000, 14840, 14680]]
# analytically-zero gradients (summation noise on both sides): compared against the global scale
ZERO_GRADS = ("depthwise_conv.bias", "linear_k.bias")


def _model(over, vocab):
    from nemo_amd.models import EncDecCTCModel, conformer_ctc_config
    cfg = conformer_ctc_config("small", vocab_size=vocab, **over)
    cfg["preprocessor"]["dither"] = 0.0
    return EncDecCTCModel(cfg)


def _load(model, P):
    sd = {k: v.detach().clone() for k, v in P.items() if k.startswith(("encoder.", "decoder."))}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.startswith("preprocessor.") for m in missing)


def _grads(model):
    return {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters()}


def _worst(got, ref, floor):
    gmax = max(r.abs().max().item() for r in ref.values())
    worst = ("", 0.0)
    for k, r in ref.items():
        r = torch.as_tensor(r).float()
        s = max(r.abs().max().item(), floor)
        if k.endswith(ZERO_GRADS):
            s = max(s, 1e-2 * gmax)
        e = (got[k] - r).abs().max().item() / s
        if e > worst[1]:
            worst = (k, e)
    return worst


def _hip_step(P, over, vocab, batch, packed, dtype=torch.float32, host_lengths=True):
    model = _model(dict(over, compute_dtype=dtype), vocab)
    model.decoder.compute_dtype = dtype
    _load(model, P)
    model = model.to(dev).train()
    enc = model.encoder
    enc.use_graphs = False
    enc.packed_rows = True if packed else False
    for fp in model.flats():
        fp.zero_grad()
    # whatever the allocator hands out next is NaN: a packed kernel that reads a frame beyond its utterance (or multiplies a zero by
    # a statistic nobody wrote) shows up as a NaN gradient instead of passing on a lucky zero page
    junk = [torch.full((32 << 20,), float("nan"), device=dev) for _ in range(4)]
    del junk
    out = model.training_step([t.to(dev) for t in batch])
    out["loss"].backward()
    torch.cuda.synchronize()
    return out["loss"].item(), _grads(model), enc.packed_last, model


@pytest.mark.parametrize("lengths", LENGTHS)
def test_packed_chain_equals_the_padded_chain_and_the_packed_oracle_fp32(golden_dir, lengths):
    from oracle import packed_ref as PK
    z = np.load(os.path.join(golden_dir, "ref_tiny_model.npz"))
    cfg = R.ConformerCfg(d_model=32, n_heads=4, n_layers=2, vocab=16, dropout=0, dropout_att=0, dropout_pre_encoder=0)
    audio, tok = torch.from_numpy(z["audio"]), torch.from_numpy(z["tokens"])
    alen, tl = torch.tensor(lengths), torch.tensor([3, 2, 3])
    # ---- oracle: the layers on the valid frames only (oracle/packed_ref.py)
    P = {k[2:]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith("P/")}
    for k in R.trainable_keys(P):
        P[k].requires_grad_(True)
    with torch.no_grad():
        mel, mel_len = R.log_mel_features(audio, alen, n_mels=cfg.feat_in)
    enc, enc_len = PK.encoder_forward_packed(P, cfg, mel, mel_len, bn_training=True, pfx="encoder.")
    logp = R.decoder_forward(P, enc, "decoder.decoder_layers.0.", cfg)
    loss, _ = R.ctc_loss_mean_batch(logp, tok, enc_len, tl, cfg.vocab)
    loss.backward()
    ref_g = {k: P[k].grad.detach().clone() for k in R.trainable_keys(P)}
    # ---- the HIP path twice
    P0 = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("P/")}
    over = dict(d_model=32, n_heads=4, n_layers=2, dropout=0.0, dropout_pre_encoder=0.0, dropout_att=0.0)
    batch = [audio, alen, tok, tl]
    l_pad, g_pad, pl_pad, _ = _hip_step(P0, over, 16, batch, packed=False)
    l_pk, g_pk, pl_pk, _ = _hip_step(P0, over, 16, batch, packed=True)
    assert pl_pad is None
    n_valid, T2 = int(enc_len.sum()), int(enc.shape[-1])
    assert pl_pk == (n_valid, 3 * T2), (pl_pk, n_valid, T2)   # the chain really ran on the valid frames only
    assert abs(l_pk - l_pad) <= 1e-4 * abs(l_pad) and abs(l_pk - loss.item()) <= 1e-3 * abs(loss.item()), (l_pk, l_pad, loss.item())
    w = _worst(g_pk, g_pad, floor=1e-3)
    assert w[1] < 1e-3, ("packed vs padded HIP", w)
    w = _worst(g_pk, ref_g, floor=1e-3)
    assert w[1] < 2e-3, ("packed HIP vs packed oracle", w)


def test_packed_chain_bf16_fused_attention_with_row_offsets():
    """the production kernels: bf16, d_k = 64 (fused rel-pos attention addressing utterance b at row_offsets[b]), grouped weight
    gradients on the side stream, layer-boundary LayerNorm pairs.  Packed vs padded on the same weights and a ragged batch whose
    shortest utterance ends inside the first key tile: loss to 1e-3, every large gradient tensor to bf16 accuracy (cos > 0.999)."""
    cfg = R.ConformerCfg(d_model=256, n_heads=4, n_layers=3, vocab=20, dropout=0, dropout_att=0, dropout_pre_encoder=0)
    P = R.init_params(cfg, seed=11)
    audio, alen, tok, tl = R.synthetic_batch(5, 3.0, vocab=20, seed=23)
    alen = torch.tensor([48000, 30000, 47000, 1600, 21000])
    tl = torch.minimum(tl, torch.tensor([8, 6, 8, 1, 4]))
    over = dict(d_model=256, n_heads=4, n_layers=3, dropout=0.0, dropout_pre_encoder=0.0, dropout_att=0.0)
    batch = [audio, alen, tok, tl]
    l_pad, g_pad, _, _ = _hip_step(P, over, 20, batch, packed=False, dtype=torch.bfloat16)
    l_pk, g_pk, pl, _ = _hip_step(P, over, 20, batch, packed=True, dtype=torch.bfloat16)
    assert pl is not None and pl[0] < 0.75 * pl[1], pl
    assert abs(l_pk - l_pad) <= 2e-3 * abs(l_pad), (l_pk, l_pad)
    n = 0
    for k, r in g_pad.items():
        if r.numel() < 256 or r.norm() < 1e-4 or k.endswith(ZERO_GRADS):
            continue
        cos = torch.dot(g_pk[k].flatten(), r.flatten()) / (g_pk[k].norm() * r.norm() + 1e-20)
        assert cos > 0.999, (k, cos.item())
        assert abs(g_pk[k].norm() / r.norm() - 1) < 2e-2, (k, g_pk[k].norm().item(), r.norm().item())
        n += 1
    assert n > 40, n


def test_auto_mode_packs_only_with_host_lengths_and_enough_padding():
    """"auto" (the default): no device sync ever -- the batch is packed only if the caller attached the lengths on the host
    (`length.host_lengths`, what the input pipeline does) and at least `packed_min_padding` of the frames are padding"""
    cfg = R.ConformerCfg(d_model=64, n_heads=4, n_layers=2, vocab=20, dropout=0, dropout_att=0, dropout_pre_encoder=0)
    P = R.init_params(cfg, seed=5)
    audio, alen, tok, tl = R.synthetic_batch(3, 1.3, vocab=20, seed=99)
    alen = torch.tensor([20800, 15000, 7777]); tl = torch.tensor([3, 2, 3])
    over = dict(d_model=64, n_heads=4, n_layers=2, dropout=0.0, dropout_pre_encoder=0.0, dropout_att=0.0)
    model = _model(over, 20)
    _load(model, P)
    model = model.to(dev).train()
    enc = model.encoder
    assert enc.packed_rows == "auto"
    b = [audio.to(dev), alen.to(dev), tok.to(dev), tl.to(dev)]
    l0 = model.training_step(b)["loss"].item()
    assert enc.packed_last is None                      # lengths only on the device: padded path, no sync
    b[1].host_lengths = alen.clone()
    l1 = model.training_step(b)["loss"].item()
    assert enc.packed_last is not None and enc.packed_last[0] < enc.packed_last[1]
    assert abs(l1 - l0) <= 1e-4 * abs(l0)
    full = torch.full((3,), audio.shape[1])
    b2 = [audio.to(dev), full.to(dev), tok.to(dev), tl.to(dev)]
    b2[1].host_lengths = full.clone()
    model.training_step(b2)
    assert enc.packed_last is None                      # nothing to skip


@pytest.mark.parametrize("v4", [1, 0])
def test_relu_mask_gemm_skips_tiles_beyond_the_utterance(v4):
    """the sub-sampling convolutions' GEMM epilogue zeroes rows beyond their utterance's length (EPI_RELU_MASK); a 256-row tile made
    of such rows only is now written as zeros without running its K loop (gemm_bf16_v4_kernel).  Against relu(A W^T + b) * mask
    in fp32 on bf16-rounded operands: utterances that end inside a tile, on a tile edge, at zero length and at full length;
    poisoned output buffer (a skipped tile that forgot to write shows as NaN)."""
    from nemo_amd import ops as o
    prev = o.gemm_config(4, v4)
    try:
        g = torch.Generator().manual_seed(3)
        B, T, F_in, N, K = 6, 1040, 16, 256, 128     # rows (b, t, f): 16 640 per utterance = 65 tiles of 256 rows
        M = B * T * F_in
        lens = torch.tensor([1040, 513, 0, 512, 37, 1039])
        A = torch.randn(M, K, generator=g).bfloat16()
        W = (torch.randn(N, K, generator=g) * 0.1).bfloat16()
        bias = torch.randn(N, generator=g)
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        o.gemm(A.to(dev), W.to(dev), out, M, N, K, K, K, N, bias=bias.to(dev), epi=o.EPI_RELU_MASK, row_len=lens.to(dev),
               rows_per_b=T * F_in, rows_inner=F_in)
        torch.cuda.synchronize()
        ref = torch.relu(A.float().to(dev) @ W.float().to(dev).t() + bias.to(dev))
        t_of = (torch.arange(M, device=dev) % (T * F_in)) // F_in
        b_of = torch.arange(M, device=dev) // (T * F_in)
        ref = ref * (t_of < lens.to(dev)[b_of]).unsqueeze(1)
        got = out.float()
        assert torch.isfinite(got).all()
        assert (got - ref).abs().max() <= 2e-2 * ref.abs().max()
        assert (got[ref == 0] == 0).all()
    finally:
        o.gemm_config(4, prev)
