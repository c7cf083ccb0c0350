"""-m gpu: the ConformerEncoder / EncDecCTCModel options beyond the five BASELINE configurations, on the HIP path, against fixtures
the reference's own classes produced (tests/golden/ref_encoder_options.npz, ref_encoder_structure.npz; oracle/make_golden.py --
the oracle restatements of the same options are pinned to the same fixtures in tests/test_oracle_pinning.py): limited / chunked
attention context, InterCTC, feat_out projection, stochastic depth.  Tiny fp32 models (north_star tolerance 1e-3 relative)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

dev = "cuda"
OPTION_GRADS = ["encoder.pre_encode.out.weight", "encoder.layers.0.self_attn.linear_q.weight", "encoder.layers.0.self_attn.pos_bias_u",
                "encoder.layers.0.conv.depthwise_conv.weight", "encoder.layers.0.conv.batch_norm.weight",
                "encoder.layers.1.feed_forward2.linear2.weight", "encoder.layers.1.norm_out.weight",
                "decoder.decoder_layers.0.weight", "decoder.decoder_layers.0.bias"]


def _model(cfg_over, vocab, **kw):
    from nemo_amd.models import EncDecCTCModel, conformer_ctc_config
    cfg = conformer_ctc_config("small", vocab_size=vocab, **cfg_over)
    cfg["preprocessor"]["dither"] = 0.0
    cfg.update(kw)
    return EncDecCTCModel(cfg)


def _tiny_with_options(golden_dir, case, enc_kw, **model_kw):
    z = np.load(os.path.join(golden_dir, "ref_tiny_model.npz"))
    zo = np.load(os.path.join(golden_dir, "ref_encoder_options.npz"))
    P = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("P/")}
    for k in zo.files:  # the parameters that differ from the tiny fixture (9-tap depthwise kernels, ...)
        if k.startswith(case + "/P/"):
            P[k[len(case) + 3:]] = torch.from_numpy(zo[k])
    over = dict(d_model=32, n_heads=4, n_layers=2, dropout=0.0, dropout_pre_encoder=0.0, dropout_att=0.0, conv_kernel_size=9, **enc_kw)
    model = _model(over, vocab=16, **model_kw)
    sd = {k: v.clone() for k, v in P.items() if k.startswith(("encoder.", "decoder."))}
    if enc_kw.get("conv_norm_type") == "layer_norm":  # a LayerNorm has no running statistics (the tiny fixture's BatchNorm buffers)
        sd = {k: v for k, v in sd.items() if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.startswith("preprocessor.") for m in missing), (missing, unexpected)
    model = model.to(dev).train()
    batch = [torch.from_numpy(z[k]).to(dev) for k in ("audio", "audio_len", "tokens", "token_len")]
    return model, batch, zo


def _check_grads(model, zo, case):
    got = {n: p.grad.detach().float().cpu().numpy() for n, p in model.named_parameters() if p.grad is not None}
    for k in OPTION_GRADS:
        ref = zo[f"{case}/grad/{k}"]
        scale = max(np.abs(ref).max(), 1e-4)
        assert np.abs(got[k] - ref).max() <= 2e-3 * scale + 1e-5, (k, np.abs(got[k] - ref).max(), scale)


@pytest.mark.parametrize("case,kw", [
    ("att_regular_8_4", dict(att_context_size=[8, 4], att_context_style="regular")),
    ("att_regular_left_6", dict(att_context_size=[6, -1], att_context_style="regular")),
    ("att_chunked_8_3", dict(att_context_size=[8, 3], att_context_style="chunked_limited")),
])
def test_limited_attention_context_matches_the_reference_fixture(golden_dir, case, kw):
    """att_context_size / att_context_style (conformer_encoder.py:794-823): the window is a mask inside the softmax kernel of the
    GEMM + softmax attention path (mi355x_relpos_softmax_fwd_ctx); encoder output, loss and gradients from every block"""
    model, batch, zo = _tiny_with_options(golden_dir, case, kw)
    for fp in model.flats():
        fp.zero_grad()
    out = model.training_step(batch)
    out["loss"].backward()
    mel, mel_len = model.preprocessor(input_signal=batch[0], length=batch[1])
    with torch.no_grad():
        enc, enc_len = model.encoder(audio_signal=mel, length=mel_len)
    torch.cuda.synchronize()
    ref_loss = float(zo[f"{case}/loss"])
    assert abs(out["loss"].item() - ref_loss) <= 1e-3 * abs(ref_loss), (out["loss"].item(), ref_loss)
    assert np.array_equal(enc_len.cpu().numpy(), zo[f"{case}/enc_len"])
    assert np.abs(enc.detach().cpu().numpy() - zo[f"{case}/enc"]).max() < 2e-3
    _check_grads(model, zo, case)


@pytest.mark.parametrize("case,kw,inter", [
    ("conv_layer_norm", dict(conv_norm_type="layer_norm"), None),
    ("conv_causal", dict(conv_context_size="causal"), None),
    ("conv_context_6_2", dict(conv_context_size=[6, 2]), None),
    ("streaming_recipe", dict(att_context_size=[8, 3], att_context_style="chunked_limited", conv_context_size="causal",
                              conv_norm_type="layer_norm"), ([1], [0.25])),
])
def test_conv_module_options_match_the_reference_fixture(golden_dir, case, kw, inter):
    """conv_norm_type = layer_norm (conformer_modules.py:293-306, 335-340), conv_context_size = 'causal' / [6, 2] (CausalConv1D,
    causal_convs.py:89-150) and the cache-aware streaming recipe's combination of chunked attention, causal LayerNorm conv module and
    InterCTC: encoder output, loss (and its InterCTC parts) and gradients from every block"""
    mkw = dict(interctc=dict(apply_at_layers=inter[0], loss_weights=inter[1])) if inter else {}
    model, batch, zo = _tiny_with_options(golden_dir, case, kw, **mkw)
    for fp in model.flats():
        fp.zero_grad()
    out = model.training_step(batch)
    out["loss"].backward()
    mel, mel_len = model.preprocessor(input_signal=batch[0], length=batch[1])
    with torch.no_grad():
        enc, enc_len = model.encoder(audio_signal=mel, length=mel_len)
    torch.cuda.synchronize()
    ref_loss = float(zo[f"{case}/loss"])
    assert abs(out["loss"].item() - ref_loss) <= 1e-3 * abs(ref_loss), (out["loss"].item(), ref_loss)
    if inter:
        for l in inter[0]:
            ref = float(zo[f"{case}/inter_ctc_loss_l{l}"])
            assert abs(out["log"][f"inter_ctc_loss_l{l}"].item() - ref) <= 1e-3 * abs(ref)
    assert np.array_equal(enc_len.cpu().numpy(), zo[f"{case}/enc_len"])
    assert np.abs(enc.detach().cpu().numpy() - zo[f"{case}/enc"]).max() < 2e-3
    _check_grads(model, zo, case)


def test_several_attention_contexts_are_drawn_per_training_step_and_the_first_one_evaluates():
    """att_context_size = [[8, 3], [-1, -1]] with att_context_probs (conformer_encoder.py:620-625): training draws, evaluation uses
    the first window"""
    from nemo_amd.modules.conformer_encoder import ConformerEncoder
    enc = ConformerEncoder(feat_in=80, n_layers=1, d_model=32, n_heads=4, conv_kernel_size=9, att_context_size=[[8, 3], [-1, -1]],
                           att_context_style="chunked_limited", att_context_probs=[0.5, 0.5])
    enc.train()
    seen = set()
    for _ in range(64):
        enc._pick_ctx()
        seen.add(enc._ctx)
    assert seen == {(2, 8, 3), (0, -1, -1)}
    enc.eval()
    enc._pick_ctx()
    assert enc._ctx == (2, 8, 3)
    with pytest.raises(ValueError):
        ConformerEncoder(feat_in=80, n_layers=1, d_model=32, n_heads=4, att_context_size=[7, 3], att_context_style="chunked_limited")


def test_interctc_losses_and_gradients_match_the_reference_fixture(golden_dir):
    """interctc: {apply_at_layers: [0, 1], loss_weights: [0.3, 0.1]} (parts/mixins/interctc_mixin.py:214-270): the captured layer
    outputs, every part of the loss and gradients from every block -- the captures' gradients join the backward chain at their layers"""
    case = "interctc_l0_l1"
    model, batch, zo = _tiny_with_options(golden_dir, case, {}, interctc=dict(loss_weights=[0.3, 0.1], apply_at_layers=[0, 1]))
    for fp in model.flats():
        fp.zero_grad()
    out = model.training_step(batch)
    caps = {l: t.detach().cpu().numpy() for l, t in model.encoder.captured.items()}
    out["loss"].backward()
    torch.cuda.synchronize()
    for key in ("loss",):
        ref = float(zo[f"{case}/{key}"])
        assert abs(out["loss"].item() - ref) <= 1e-3 * abs(ref), (key, out["loss"].item(), ref)
    for key in ("final_loss", "inter_ctc_loss_l0", "inter_ctc_loss_l1"):
        ref = float(zo[f"{case}/{key}"])
        assert abs(out["log"][key].item() - ref) <= 1e-3 * abs(ref), (key, out["log"][key].item(), ref)
    for l in (0, 1):
        assert np.abs(caps[l] - zo[f"{case}/layer_output_{l}"]).max() < 2e-3, l
    _check_grads(model, zo, case)


# gradients that are analytically ZERO (depthwise bias under batch-statistics BatchNorm; key bias under softmax shift invariance):
# both sides hold pure summation-rounding noise, so they are compared absolutely against the global scale (tests/test_model_gpu.py)
ZERO_GRADS = ("depthwise_conv.bias", "linear_k.bias")


def _cmp_probe_grads(z, case, got):
    refs = {k[len(case) + 6:]: z[k] for k in z.files if k.startswith(f"{case}/grad/")}
    gmax = max(np.abs(r).max() for r in refs.values())
    for name, ref in refs.items():
        scale = max(np.abs(ref).max(), 1e-4)
        if name.endswith(ZERO_GRADS):
            scale = max(scale, 1e-2 * gmax)
        assert np.abs(got[name] - ref).max() <= 2e-3 * scale + 1e-5, (name, np.abs(got[name] - ref).max(), scale)
    return len(refs)


def _structure_case(golden_dir, case, pname=None):
    z = np.load(os.path.join(golden_dir, "ref_encoder_structure.npz"))
    f32 = lambda a: torch.from_numpy(a.astype(np.float32) if a.dtype == np.float16 else a)
    pname = pname or case
    P = {k[len(pname) + 3:]: f32(z[k]) for k in z.files if k.startswith(pname + "/P/")}
    return z, P, f32


def test_feat_out_projection_matches_the_reference_fixture(golden_dir):
    """feat_out = 24 on d_model = 32 (conformer_encoder.py:474-479, 738-739): output, lengths, and the gradients of a random probe
    for layer 0, the projection itself and the sub-sampling's first convolution"""
    from nemo_amd.modules.conformer_encoder import ConformerEncoder
    z, P, f32 = _structure_case(golden_dir, "feat_out")
    enc = ConformerEncoder(feat_in=80, n_layers=2, d_model=32, n_heads=4, feat_out=24, conv_kernel_size=9, dropout=0.0,
                           dropout_pre_encoder=0.0, dropout_emb=0.0, dropout_att=0.0, compute_dtype=torch.float32)
    missing, unexpected = enc.load_state_dict(P, strict=False)
    assert not unexpected and not [m for m in missing if "pos_enc" not in m], (missing, unexpected)
    enc = enc.to(dev).train()
    x, n = f32(z["feat_out/x"]).to(dev), torch.from_numpy(z["feat_out/len"]).to(dev)
    y, yl = enc(audio_signal=x, length=n)
    (y * f32(z["feat_out/probe"]).to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert tuple(y.shape) == tuple(z["feat_out/y"].shape) and np.array_equal(yl.cpu().numpy(), z["feat_out/ylen"])
    assert np.abs(y.detach().cpu().numpy() - z["feat_out/y"]).max() < 2e-3
    got = {k: p.grad.detach().cpu().numpy() for k, p in enc.named_parameters() if p.grad is not None}
    n_checked = _cmp_probe_grads(z, "feat_out", got)
    assert n_checked > 10 and "out_proj.weight" in got


def _encoder_probe_case(golden_dir, case, **enc_kw):
    """an encoder of the ref_encoder_structure.npz geometry with the fixture's parameters: output, lengths, probe gradients"""
    from nemo_amd.modules.conformer_encoder import ConformerEncoder
    z, P, f32 = _structure_case(golden_dir, case)
    kw = dict(feat_in=80, n_layers=2, d_model=32, n_heads=4, conv_kernel_size=9, dropout=0.0, dropout_pre_encoder=0.0,
              dropout_emb=0.0, dropout_att=0.0, compute_dtype=torch.float32)
    kw.update(enc_kw)
    enc = ConformerEncoder(**kw)
    missing, unexpected = enc.load_state_dict(P, strict=False)
    assert not unexpected and not [m for m in missing if "pos_enc" not in m], (missing, unexpected)
    enc = enc.to(dev).train()
    x, n = f32(z[f"{case}/x"]).to(dev), torch.from_numpy(z[f"{case}/len"]).to(dev)
    y, yl = enc(audio_signal=x, length=n)
    (y * f32(z[f"{case}/probe"]).to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert tuple(y.shape) == tuple(z[f"{case}/y"].shape) and np.array_equal(yl.cpu().numpy(), z[f"{case}/ylen"])
    assert np.abs(y.detach().cpu().numpy() - z[f"{case}/y"]).max() < 2e-3
    got = {k: p.grad.detach().cpu().numpy() for k, p in enc.named_parameters() if p.grad is not None}
    assert _cmp_probe_grads(z, case, got) > 10
    return enc


def test_local_attention_matches_the_reference_fixture(golden_dir):
    """self_attention_model = rel_pos_local_attn, window [6, 6] (RelPositionMultiHeadAttentionLongformer, the long-form recipes):
    ragged lengths -- a sequence shorter than the window, one that is no multiple of 2w, padded queries -- against the reference
    class's own output and gradients.  Runs as the banded 'regular' context of the rel_pos model (same positional rows)."""
    enc = _encoder_probe_case(golden_dir, "local_attn", self_attention_model="rel_pos_local_attn", att_context_size=[6, 6])
    assert enc._ctx_limited_any()
    with pytest.raises(ValueError):
        type(enc)(feat_in=80, n_layers=1, d_model=32, n_heads=4, self_attention_model="rel_pos_local_attn")
    with pytest.raises(NotImplementedError):
        type(enc)(feat_in=80, n_layers=1, d_model=32, n_heads=4, self_attention_model="rel_pos_local_attn", att_context_size=[6, 4])


def test_causal_downsampling_striding_matches_the_reference_fixture(golden_dir):
    """causal_downsampling = True on the 'striding' x4 stack (CausalConv2D, causal_convs.py:24-72: F.pad (2, 1) on time and frequency,
    no symmetric padding): another sampling grid (101 frames -> 51 -> 26, 80 bins -> 41 -> 21), ragged lengths; conv1's direct
    kernel and the im2col / col2im pair run with pad = 2"""
    enc = _encoder_probe_case(golden_dir, "causal_striding", causal_downsampling=True)
    assert enc.pre_encode._pad == 2 and enc.pre_encode._feat_after == 21


def test_causal_downsampling_streaming_fastconformer_matches_the_reference_fixture(golden_dir):
    """the cache-aware streaming recipe's whole encoder combination (conf/fastconformer/cache_aware_streaming/*.yaml): dw_striding x8
    with CausalConv2D stages (depthwise 3x3 stride-2 kernels with pad = 2), chunked_limited attention [8, 3], causal LayerNorm conv
    module -- against the reference encoder's own output and gradients"""
    enc = _encoder_probe_case(golden_dir, "streaming_fastconformer", subsampling="dw_striding", subsampling_factor=8,
                              subsampling_conv_channels=16, causal_downsampling=True, att_context_size=[8, 3],
                              att_context_style="chunked_limited", conv_context_size="causal", conv_norm_type="layer_norm")
    assert enc.pre_encode._pad == 2 and enc.pre_encode._feat_after == 11 and enc._ctx_limited_any()


@pytest.mark.parametrize("sub", ["striding", "dw_striding"])
def test_causal_downsampling_bf16_production_paths_follow_fp32(sub):
    """causal_downsampling at a width where bf16 takes the production kernels -- 'striding': conv2 as an implicit GEMM whose gather
    taps are (k - 2), the input gradient's parity classes swapped (the tiny fp32 fixtures run im2col / col2im); 'dw_striding': the
    bf16 depthwise kernels -- against the SAME encoder computing in fp32 (pinned to the reference by the fixture tests above)"""
    from nemo_amd.modules.conformer_encoder import ConformerEncoder
    kw = dict(feat_in=80, n_layers=1, d_model=256, n_heads=4, conv_kernel_size=9, causal_downsampling=True, dropout=0.0,
              dropout_pre_encoder=0.0, dropout_emb=0.0, dropout_att=0.0, subsampling=sub,
              subsampling_factor=4 if sub == "striding" else 8)
    torch.manual_seed(5)
    e32 = ConformerEncoder(compute_dtype=torch.float32, **kw).to(dev).train()
    e16 = ConformerEncoder(compute_dtype=torch.bfloat16, **kw).to(dev).train()
    e16.load_state_dict(e32.state_dict())
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 80, 203, generator=g).to(dev); n = torch.tensor([203, 150]).to(dev)
    probe = None
    outs, grads = [], []
    for e in (e32, e16):
        y, yl = e(audio_signal=x, length=n)
        if probe is None:
            probe = torch.randn(y.shape, generator=g).to(dev)
        (y * probe).sum().backward()
        torch.cuda.synchronize()
        outs.append((y.detach().float().cpu(), yl.cpu()))
        grads.append({k: p.grad.detach().float().cpu() for k, p in e.named_parameters() if k.startswith("pre_encode.") and p.grad is not None})
    assert torch.equal(outs[0][1], outs[1][1])
    rel = lambda a, b: ((a - b).norm() / (b.norm() + 1e-12)).item()
    assert rel(outs[1][0], outs[0][0]) < 3e-2, rel(outs[1][0], outs[0][0])
    assert len(grads[0]) >= 6
    for k, g32 in grads[0].items():
        assert rel(grads[1][k], g32) < 0.12, (k, rel(grads[1][k], g32))   # (bf16 through three ReLU stages; a wrong tap is O(1))


def test_cache_aware_streaming_recipe_geometry_trains_in_bf16():
    """the cache-aware streaming FastConformer-CTC recipe's encoder combination (dw_striding x8 with causal down-sampling,
    chunked_limited attention, causal LayerNorm conv module, un-normalised features) as a model on the bf16 production kernels: a
    few optimizer steps on one batch -- finite, decreasing loss, every parameter receives a gradient"""
    from nemo_amd.models import EncDecCTCModel, conformer_ctc_config
    cfg = conformer_ctc_config("small", vocab_size=32, d_model=128, n_heads=4, n_layers=2, subsampling="dw_striding",
                               subsampling_factor=8, subsampling_conv_channels=128, causal_downsampling=True,
                               att_context_size=[16, 3], att_context_style="chunked_limited", conv_kernel_size=9,
                               conv_context_size="causal", conv_norm_type="layer_norm", dropout=0.0, dropout_pre_encoder=0.0,
                               dropout_att=0.0, compute_dtype=torch.bfloat16)
    cfg["preprocessor"].update(dither=0.0, normalize="NA")
    torch.manual_seed(3)
    m = EncDecCTCModel(cfg)
    m.decoder.compute_dtype = torch.bfloat16
    m = m.to(dev).train()
    m.setup_optimization(dict(name="adamw", lr=2e-3, betas=[0.9, 0.98], weight_decay=0.0))
    g = torch.Generator().manual_seed(4)
    audio = (0.1 * torch.randn(3, 48000, generator=g)).to(dev)
    alen = torch.tensor([48000, 36000, 20000]).to(dev)
    tok = torch.randint(0, 32, (3, 12), generator=g).to(dev); tl = torch.tensor([12, 9, 5]).to(dev)
    losses = [m.fit_step([audio, alen, tok, tl])["loss"].item() for _ in range(8)]
    torch.cuda.synchronize()
    assert all(np.isfinite(losses)) and losses[-1] < 0.8 * losses[0], losses
    assert m.encoder.pre_encode._pad == 2 and m.preprocessor.featurizer.normalize == "NA"


def test_bypass_pre_encode_with_layer_norm_conv_module_matches_the_reference_fixture(golden_dir):
    """the reference's own encoder test geometry (tests/collections/asr/test_conformer_encoder.py:129-199): pre-encoded frames
    [B, T, d_model = 16] through three layers with a LayerNorm conv module of kernel 3 and a feat_out = 8 projection"""
    from nemo_amd.modules.conformer_encoder import ConformerEncoder
    z, P, f32 = _structure_case(golden_dir, "bypass")
    enc = ConformerEncoder(feat_in=10, n_layers=3, d_model=16, feat_out=8, stochastic_depth_drop_prob=0.0, dropout=0.0,
                           dropout_pre_encoder=0.0, dropout_emb=0.0, dropout_att=0.0, conv_norm_type="layer_norm", conv_kernel_size=3,
                           compute_dtype=torch.float32)
    missing, unexpected = enc.load_state_dict(P, strict=False)
    assert not unexpected and not [m for m in missing if "pos_enc" not in m], (missing, unexpected)
    enc = enc.to(dev).train()
    x, n = f32(z["bypass/x"]).to(dev), torch.from_numpy(z["bypass/len"]).to(dev)
    y, yl = enc(audio_signal=x, length=n, bypass_pre_encode=True)
    (y * f32(z["bypass/probe"]).to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert tuple(y.shape) == tuple(z["bypass/y"].shape) and np.array_equal(yl.cpu().numpy(), z["bypass/ylen"])
    assert np.abs(y.detach().cpu().numpy() - z["bypass/y"]).max() < 2e-3
    got = {k: p.grad.detach().cpu().numpy() for k, p in enc.named_parameters() if p.grad is not None}
    assert _cmp_probe_grads(z, "bypass", got) > 10
    with pytest.raises(ValueError, match="bypass_pre_encode is True"):
        enc(audio_signal=x.transpose(1, 2), length=n, bypass_pre_encode=True)
    with pytest.raises(ValueError, match="bypass_pre_encode is False"):
        enc(audio_signal=x, length=n)


@pytest.mark.parametrize("mode", ["uniform", "linear"])
def test_stochastic_depth_draws_the_reference_decisions_and_matches_its_fixture(golden_dir, mode):
    """stochastic_depth_drop_prob = 0.6 over 4 layers (conformer_encoder.py:696-707): one torch.rand(1) per droppable layer from the
    global generator at the reference's point of the sequence -- same seed, same dropped layers, same output and gradients (a dropped
    layer contributes x * 0, a kept one is rescaled by 1 / (1 - p))"""
    from nemo_amd.modules.conformer_encoder import ConformerEncoder
    case = f"sd_{mode}"
    z, P, f32 = _structure_case(golden_dir, case, pname="sd")
    enc = ConformerEncoder(feat_in=80, n_layers=4, d_model=32, n_heads=4, conv_kernel_size=9, dropout=0.0, dropout_pre_encoder=0.0,
                           dropout_emb=0.0, dropout_att=0.0, stochastic_depth_drop_prob=0.6, stochastic_depth_mode=mode,
                           stochastic_depth_start_layer=1, compute_dtype=torch.float32)
    assert np.allclose(enc.layer_drop_probs, z[f"{case}/probs"])
    missing, unexpected = enc.load_state_dict(P, strict=False)
    assert not unexpected and not [m for m in missing if "pos_enc" not in m], (missing, unexpected)
    enc = enc.to(dev).train()
    x, n = f32(z[f"{case}/x"]).to(dev), torch.from_numpy(z[f"{case}/len"]).to(dev)
    torch.manual_seed(int(z[f"{case}/seed"]))
    y, yl = enc(audio_signal=x, length=n)
    (y * f32(z[f"{case}/probe"]).to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert np.array_equal(yl.cpu().numpy(), z[f"{case}/ylen"])
    assert np.abs(y.detach().cpu().numpy() - z[f"{case}/y"]).max() < 2e-3
    got = {k: p.grad.detach().cpu().numpy() for k, p in enc.named_parameters() if p.grad is not None}
    _cmp_probe_grads(z, case, got)
    # evaluation: no layer is dropped, nothing is rescaled
    enc.eval()
    with torch.no_grad():
        y0, _ = enc(audio_signal=x, length=n)
        enc2 = enc
        enc2.layer_drop_probs = [0.0] * 4
        y1, _ = enc2(audio_signal=x, length=n)
    assert torch.equal(y0, y1)


def test_skip_nan_grad_zeroes_the_gradients_of_a_poisoned_step(golden_dir):
    """skip_nan_grad: true (models/asr_model.py:147-174): a step whose gradients hold NaN / Inf is skipped -- with weight decay 0 the
    parameters do not move; the next clean step trains"""
    model, batch, _ = _tiny_with_options(golden_dir, "att_regular_8_4", {}, skip_nan_grad=True)
    model.setup_optimization(dict(name="adamw", lr=1e-3, betas=[0.9, 0.98], weight_decay=0.0))
    model.fit_step(batch)
    before = [fp.flat.clone() for fp in model.flats()]
    bad = [t.clone() for t in batch]
    bad[0][0, 100] = float("nan")
    model.fit_step(bad)
    assert model.skipped_steps == 1
    for b, fp in zip(before, model.flats()):
        assert torch.equal(b, fp.flat)
    out = model.fit_step(batch)
    assert model.skipped_steps == 1 and torch.isfinite(out["loss"])
    assert any(not torch.equal(b, fp.flat) for b, fp in zip(before, model.flats()))
