"""-m gpu: the drop-in modules end to end (preprocessor -> encoder -> decoder -> CTC loss, forward AND backward through
the C ABI) against (a) fixtures produced by the reference's own source files (tests/golden/ref_tiny_model.npz,
oracle/make_golden.py) and (b) the CPU oracle on fresh seeded inputs.  north_star tolerance: 1e-3 relative fp32."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import conformer_ref as R

dev = "cuda"


def _model(cfg_over, vocab, size="small", **kw):
    from nemo_amd.models import EncDecCTCModel, conformer_ctc_config
    cfg = conformer_ctc_config(size, vocab_size=vocab, **cfg_over)
    cfg["preprocessor"]["dither"] = 0.0
    cfg.update(kw)
    return EncDecCTCModel(cfg)


def _load(model, P):
    sd = {}
    for k, v in P.items():
        if k.startswith("encoder.") or k.startswith("decoder."):
            sd[k] = v.detach().clone()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(m.startswith("preprocessor.") for m in missing), missing


def _grads(model):
    g = {}
    for n, p in model.named_parameters():
        g[n] = p.grad.detach().float().cpu().clone()
    return g


# gradients that are analytically ZERO (depthwise bias under batch-statistics BatchNorm; key bias under softmax shift
# invariance): both sides hold pure summation-rounding noise, so they are compared absolutely against the global scale
ZERO_GRADS = ("depthwise_conv.bias", "linear_k.bias")


def _cmp_grads(got, ref, rtol, floor):
    """relative to each tensor's own scale, with an absolute floor for analytically-zero gradients"""
    worst = ("", 0.0)
    gmax = max(torch.as_tensor(r).abs().max().item() for r in ref.values())
    for k, r in ref.items():
        r = torch.as_tensor(r).float()
        e = (got[k] - r).abs().max().item()
        s = max(r.abs().max().item(), floor)
        if k.endswith(ZERO_GRADS):
            s = max(s, 1e-2 * gmax)
        if e / s > worst[1]:
            worst = (k, e / s)
    assert worst[1] < rtol, worst


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_tiny_model_matches_reference_fixture(golden_dir, mode):
    z = np.load(os.path.join(golden_dir, "ref_tiny_model.npz"))
    P = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("P/")}
    over = dict(d_model=32, n_heads=4, n_layers=2, dropout=0.0, dropout_pre_encoder=0.0, dropout_att=0.0)
    model = _model(over, vocab=16)
    _load(model, P)
    model = model.to(dev)
    model.train(mode == "train")
    batch = [torch.from_numpy(z[k]).to(dev) for k in ("audio", "audio_len", "tokens", "token_len")]
    for fp in model.flats():
        fp.zero_grad()
    out = model.training_step(batch)
    logp, enc_len, _ = model.forward(input_signal=batch[0], input_signal_length=batch[1]) if mode == "eval" else (None, None, None)
    loss = out["loss"]
    loss.backward()
    torch.cuda.synchronize()
    ref_loss = float(z[f"{mode}/loss"])
    assert abs(loss.item() - ref_loss) <= 1e-3 * abs(ref_loss), (loss.item(), ref_loss)
    if logp is not None:
        assert np.array_equal(enc_len.cpu().numpy(), z["eval/enc_len"])
        assert np.abs(logp.detach().cpu().numpy() - z["eval/logp"]).max() < 2e-3
    ref = {k[len(mode) + 6:]: z[k] for k in z.files if k.startswith(f"{mode}/grad/")}
    _cmp_grads(_grads(model), ref, rtol=2e-3, floor=1e-3)


def test_ragged_batch_matches_cpu_oracle_fp32():
    """d=64, 4 heads, 2 layers, B=3 ragged lengths, training-mode BatchNorm, fp32 compute: loss + every gradient."""
    cfg = R.ConformerCfg(d_model=64, n_heads=4, n_layers=2, vocab=20, dropout=0, dropout_att=0, dropout_pre_encoder=0)
    P = R.init_params(cfg, seed=5)
    audio, _, tok, _ = R.synthetic_batch(3, 1.3, vocab=20, seed=99)
    alen = torch.tensor([20800, 15000, 7777]); tl = torch.tensor([3, 2, 3])
    Pr = {k: (v.clone().requires_grad_(True) if k in R.trainable_keys(P) else v) for k, v in P.items()}
    ref = R.model_forward(Pr, cfg, audio, alen, tok, tl, train=False, bn_training=True)
    ref["loss"].backward()
    over = dict(d_model=64, n_heads=4, n_layers=2, dropout=0.0, dropout_pre_encoder=0.0, dropout_att=0.0)
    model = _model(over, vocab=20)
    _load(model, P)
    model = model.to(dev).train()
    out = model.training_step([audio.to(dev), alen.to(dev), tok.to(dev), tl.to(dev)])
    out["loss"].backward()
    torch.cuda.synchronize()
    assert abs(out["loss"].item() - ref["loss"].item()) <= 1e-3 * abs(ref["loss"].item())
    refg = {k: Pr[k].grad for k in R.trainable_keys(P)}
    _cmp_grads(_grads(model), refg, rtol=2e-3, floor=1e-3)
    # running statistics of BatchNorm follow torch (momentum 0.1, unbiased running_var)
    stats = {}
    R.model_forward(P, cfg, audio, alen, tok, tl, train=False, bn_training=True, bn_stats_out=stats)
    m0, v0 = stats["encoder.layers.0.conv."]
    bn = model.encoder.layers[0].conv.batch_norm
    assert torch.allclose(bn.running_mean.cpu(), 0.1 * m0, atol=1e-4)
    assert torch.allclose(bn.running_var.cpu(), 0.9 + 0.1 * v0, atol=1e-4)
    assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("d_model,n_heads,pad_heads", [(64, 2, True), (256, 4, True), (176, 4, True), (176, 4, False)])
def test_bf16_mfma_path_close_to_fp32_oracle(d_model, n_heads, pad_heads):
    """bf16 compute (MFMA GEMMs, bf16 activations): loss within 2 %, gradient direction cos > 0.98 per big tensor.
    d_model = 256 (d_k = 64) takes the production paths the tiny configurations skip: fused flash attention, implicit-GEMM
    conv2 (forward / weight / input gradient), grouped weight gradients on the side stream, 256x256 GEMM tiles.
    d_model = 176 / 4 heads is the recipe table's Small geometry (d_k = 44, not a multiple of 8): zero-padded heads inside the
    packed weight images, per-head batched weight gradients; there EVERY attention tensor is checked, bias-sized ones included.
    pad_heads (round 5): heads narrower than 64 are padded to the fused kernels' width and take the fused rel-pos attention
    (d_k' = 64); False = the round-4 layout (d_k' = 48, scores materialised)."""
    cfg = R.ConformerCfg(d_model=d_model, n_heads=n_heads, n_layers=2, vocab=20, dropout=0, dropout_att=0,
                         dropout_pre_encoder=0)
    P = R.init_params(cfg, seed=6)
    audio, alen, tok, tl = R.synthetic_batch(4, 1.0 if d_model == 64 else 2.5, vocab=20, seed=17)
    Pr = {k: (v.clone().requires_grad_(True) if k in R.trainable_keys(P) else v) for k, v in P.items()}
    ref = R.model_forward(Pr, cfg, audio, alen, tok, tl, train=False, bn_training=True)
    ref["loss"].backward()
    over = dict(d_model=d_model, n_heads=n_heads, n_layers=2, dropout=0.0, dropout_pre_encoder=0.0, dropout_att=0.0,
                compute_dtype=torch.bfloat16)
    model = _model(over, vocab=20)
    model.decoder.compute_dtype = torch.bfloat16
    _load(model, P)
    model = model.to(dev).train()
    model.encoder.flash_pad_heads = pad_heads
    dk = d_model // n_heads
    assert model.encoder._geometry(torch.bfloat16)[1] == (64 if (pad_heads and dk < 64) else (dk + 7) // 8 * 8)
    out = model.training_step([audio.to(dev), alen.to(dev), tok.to(dev), tl.to(dev)])
    out["loss"].backward()
    torch.cuda.synchronize()
    assert abs(out["loss"].item() - ref["loss"].item()) <= 2e-2 * abs(ref["loss"].item()), (out["loss"].item(), ref["loss"].item())
    g = _grads(model)
    n_att = 0
    for k in R.trainable_keys(P):
        r = Pr[k].grad.flatten()
        att = d_model == 176 and ".self_attn." in k and r.norm() > 1e-5
        if (r.numel() < 1024 or r.norm() < 1e-3) and not att:
            continue
        cos = torch.dot(g[k].flatten(), r) / (g[k].norm() * r.norm() + 1e-20)
        assert cos > (0.98 if r.numel() >= 1024 else 0.95), (k, cos.item())
        n_att += att
    assert d_model != 176 or n_att >= 2 * 9, n_att  # q / k / v / out weights + biases, linear_pos, pos_bias_u / v per layer


def test_dropout_training_runs_and_is_stochastic():
    over = dict(d_model=64, n_heads=4, n_layers=2)
    model = _model(over, vocab=20).to(dev).train()
    model.preprocessor.featurizer.dither = 1e-5
    audio, alen, tok, tl = R.synthetic_batch(2, 1.0, vocab=20, seed=3)
    batch = [audio.to(dev), alen.to(dev), tok.to(dev), tl.to(dev)]
    l1 = model.training_step(batch)["loss"]
    l1.backward()
    l2 = model.training_step(batch)["loss"]
    torch.cuda.synchronize()
    assert torch.isfinite(l1) and torch.isfinite(l2) and l1.item() != l2.item()
    for n, p in model.named_parameters():
        assert torch.isfinite(p.grad).all(), n


def test_spec_augment_in_training_step_matches_oracle_on_masked_features():
    """ctc_models.py:532-533: SpecAugment sits between the preprocessor and the encoder, in training mode only.  With
    dropout/dither off, the loss of a training step with augmentation equals the loss of the un-augmented model fed the
    oracle-masked features (same device seed -> same rectangles)."""
    from oracle import specaug_ref as SR
    over = dict(d_model=64, n_heads=4, n_layers=2, dropout=0.0, dropout_pre_encoder=0.0, dropout_att=0.0, dropout_emb=0.0)
    sa = {"_target_": "nemo.collections.asr.modules.SpectrogramAugmentation", "freq_masks": 2, "time_masks": 4,
          "freq_width": 20, "time_width": 0.05}
    model = _model(over, vocab=20, spec_augment=sa).to(dev)
    audio, alen, tok, tl = R.synthetic_batch(3, 1.5, vocab=20, seed=21)
    batch = [audio.to(dev), alen.to(dev), tok.to(dev), tl.to(dev)]
    model.eval()
    feats, flen = model.preprocessor(input_signal=batch[0], length=batch[1])
    lp_eval, _, _ = model.forward(input_signal=batch[0], input_signal_length=batch[1])
    lp_feat, _, _ = model.forward(processed_signal=feats, processed_signal_length=flen)
    assert torch.equal(lp_eval, lp_feat)  # no augmentation outside training
    B, F, T = feats.shape
    torch.manual_seed(77)
    rects = model.spec_augmentation.mask_rects(B, F, T, flen, feats.device)
    masked = feats.cpu()
    for r, v in rects:
        masked = SR.apply_rects(masked, r.cpu(), v)
    assert (masked != feats.cpu()).float().mean() > 0.02
    model.train()
    torch.manual_seed(77)
    l_aug = model.training_step(batch)["loss"]
    model.spec_augmentation = None
    lp, enc_len, _ = model.forward(processed_signal=masked.to(dev), processed_signal_length=flen)
    l_ref = model.loss(log_probs=lp, targets=batch[2], input_lengths=enc_len, target_lengths=batch[3])
    torch.cuda.synchronize()
    assert abs(l_aug.item() - l_ref.item()) <= 1e-5 * abs(l_ref.item()), (l_aug.item(), l_ref.item())


def test_fit_steps_reduce_loss_and_nemo_roundtrip(tmp_path):
    from nemo_amd.models import EncDecCTCModel
    over = dict(d_model=64, n_heads=4, n_layers=2, dropout=0.0, dropout_pre_encoder=0.0, dropout_att=0.0)
    model = _model(over, vocab=20).to(dev).train()
    model.setup_optimization(dict(name="adamw", lr=2e-3, betas=[0.9, 0.98], weight_decay=1e-3))
    audio, alen, tok, tl = R.synthetic_batch(4, 1.0, vocab=20, seed=8)
    batch = [audio.to(dev), alen.to(dev), tok.to(dev), tl.to(dev)]
    losses = [model.fit_step(batch)["loss"].item() for _ in range(12)]
    assert losses[-1] < 0.8 * losses[0], losses
    path = str(tmp_path / "m.nemo")
    model.eval()
    lp1, _, _ = model.forward(input_signal=batch[0], input_signal_length=batch[1])
    model.save_to(path)
    m2 = EncDecCTCModel.restore_from(path, map_location=dev).eval()
    lp2, _, _ = m2.forward(input_signal=batch[0], input_signal_length=batch[1])
    torch.cuda.synchronize()
    assert torch.allclose(lp1, lp2, atol=1e-5)
    assert set(m2.state_dict().keys()) == set(model.state_dict().keys())


def test_optimizer_behind_backward_follows_the_plain_step():
    """fit_step with the optimizer running slice by slice behind backward (weight-gradient stream) against the plain
    zero_grad -> backward -> step order: the update rule per element is the same kernel, the only difference allowed is
    the float-atomic order inside the split-K weight gradients, so the loss curves agree to 1e-3"""
    over = dict(d_model=64, n_heads=4, n_layers=3, dropout=0.0, dropout_pre_encoder=0.0, dropout_att=0.0)
    audio, alen, tok, tl = R.synthetic_batch(4, 1.0, vocab=20, seed=8)
    batch = [audio.to(dev), alen.to(dev), tok.to(dev), tl.to(dev)]
    curves = []
    for early in (True, False):
        torch.manual_seed(21)
        model = _model(over, vocab=20).to(dev).train()
        model.optimizer_in_backward = early
        model.setup_optimization(dict(name="adamw", lr=1e-3, betas=[0.9, 0.98], weight_decay=1e-3,
                                      ema=dict(enable=True, decay=0.9)))
        curves.append([model.fit_step(batch)["loss"].item() for _ in range(8)])
        assert model._optimizer.step_count == 8
        assert (model.encoder.grad_ready_hook is not None) == early
    a, b = curves
    assert a[-1] < 0.9 * a[0]
    for x, y in zip(a, b):
        assert abs(x - y) <= 1e-3 * abs(y), (a, b)


def _wav_corpus(tmp_path, n, vocab, seed=0, sr=16000):
    import json, wave
    rs = np.random.RandomState(seed)
    lines = []
    for i in range(n):
        ns = int(rs.randint(int(0.6 * sr), int(1.4 * sr)))
        x = (rs.randn(ns) * 2000).astype(np.int16)
        with wave.open(str(tmp_path / f"u{i}.wav"), "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(sr); f.writeframes(x.tobytes())
        text = "".join(rs.choice(vocab[1:], size=rs.randint(3, 9)))
        lines.append(dict(audio_filepath=f"u{i}.wav", duration=ns / sr, text=text))
    m = str(tmp_path / "train.json")
    with open(m, "w") as f:
        for ln in lines:
            f.write(json.dumps(ln) + "\n")
    return m


def test_device_batch_loader_overlapped_copies_deliver_the_host_batches():
    from nemo_amd.data import DeviceBatchLoader
    g = torch.Generator().manual_seed(0)
    host = [(torch.randn(3, 4000 + 977 * (i % 5), generator=g), torch.tensor([4000, 17, 300 + i]),
             torch.randint(0, 9, (3, 2 + i % 3), generator=g), torch.tensor([2, 1, 2])) for i in range(11)]
    burn = torch.randn(2048, 2048, device=dev)
    n = 0
    for want, got in zip(host, DeviceBatchLoader(host, dev, prefetch=2)):
        burn = burn @ burn.clamp(-1e-3, 1e-3)  # compute-stream work for the copies to overlap with
        assert all(t.is_cuda for t in got)
        for a, b in zip(want, got):
            assert torch.equal(a, b.cpu())
        n += 1
    assert n == len(host)


def test_fit_from_a_manifest_with_semi_sorted_batches(tmp_path):
    """manifest -> AudioToCharDataset -> SemiSortBatchSampler -> collate -> pinned staging + copy stream -> fit_step"""
    vocab = [" "] + list("abcdefghijklmnopqrs")
    m = _wav_corpus(tmp_path, 24, vocab)
    over = dict(d_model=64, n_heads=4, n_layers=2, dropout=0.0, dropout_pre_encoder=0.0, dropout_att=0.0)
    torch.manual_seed(3)
    model = _model(over, vocab=len(vocab), labels=vocab).to(dev).train()
    model.setup_optimization(dict(name="adamw", lr=2e-3, betas=[0.9, 0.98], weight_decay=1e-3))
    dl = model.setup_training_data(dict(manifest_filepath=m, batch_size=4, use_semi_sorted_batching=True,
                                        semi_sort_synced_rng=True, num_workers=0))
    assert len(dl) == 6
    losses = torch.stack(model.fit(max_steps=18)).tolist()  # three epochs
    assert len(losses) == 18 and all(np.isfinite(losses))
    assert np.mean(losses[-6:]) < 0.9 * np.mean(losses[:6]), losses
    assert model._cfg["train_ds"]["manifest_filepath"] == m


def _dp_model(kind, over, vocab):
    torch.manual_seed(5)
    if kind == "rnnt":
        return _rnnt_model(torch.float32, **over).to(dev).train()
    if kind == "squeezeformer":
        from nemo_amd.models import EncDecCTCModel, squeezeformer_ctc_config
        cfg = squeezeformer_ctc_config("xs", vocab_size=vocab, compute_dtype=torch.float32, **over)
        cfg["preprocessor"]["dither"] = 0.0
        return EncDecCTCModel(cfg).to(dev).train()
    return _model(over, vocab=vocab).to(dev).train()


def _dp_batch(kind, vocab):
    audio, alen, tok, tl = R.synthetic_batch(4, 1.0, vocab=vocab, seed=8)
    if kind == "rnnt":  # ragged audio and label lengths through the fused joint + loss
        tl = torch.tensor([3, 2, 3, 1])
        alen = torch.tensor([16000, 12000, 14000, 9000])
    return audio, alen, tok, tl


def _dp_worker(rank, world, port, out_dir, over, vocab, kind="ctc"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # both ranks on the one GPU; gloo stages through the host
    try:
        torch.cuda.set_device(0)
        model = _dp_model(kind, over, vocab)
        model.setup_optimization(dict(name="adamw", lr=1e-3, betas=[0.9, 0.98], weight_decay=0.0))
        audio, alen, tok, tl = _dp_batch(kind, vocab)
        sl = slice(2 * rank, 2 * rank + 2)
        batch = [audio[sl].to(dev), alen[sl].to(dev), tok[sl].to(dev), tl[sl].to(dev)]
        # what fit_step does up to the optimizer: zero_grad, forward, backward with the bucketed exchange, join
        syncs = model._grad_syncs()
        model._optimizer.zero_grad()
        loss = model.training_step(batch)["loss"]
        loss.backward()
        model._after_backward()
        scale = 1.0
        for gs in syncs:
            scale = gs.wait()
        torch.cuda.synchronize()
        grads = [fp.grad.detach().cpu() * scale for fp in model.flats()]
        bn = model.encoder.layers[0].conv.batch_norm.running_mean.detach().cpu()
        # then two complete steps (optimizer behind the exchange on the second) must leave the replicas identical
        model.fit_step(batch)
        model.optimizer_in_backward = True
        model.fit_step(batch)
        torch.cuda.synchronize()
        mbx = getattr(model.encoder, "_syncbn_mailbox", None)
        torch.save(dict(grads=grads, loss=loss.detach().cpu(), bn=bn, flat=[fp.flat.detach().cpu() for fp in model.flats()],
                        mailbox=(mbx.status() + (mbx.exchanges,)) if mbx is not None else None),
                   os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mailbox", [False, True], ids=["process-group", "mailbox"])
def test_data_parallel_two_ranks_equal_one_process_on_the_joint_batch(tmp_path, mailbox, monkeypatch):
    """(mailbox: the SyncBatchNorm statistics travel through nemo_amd.mailbox.StatsMailbox -- one kernel launch per exchange over
    hipIpc-mapped memory -- instead of the process group, MI355X_SYNCBN_MAILBOX=1.)
    the N > 1 path on real kernels (two processes share the GPU, gloo carries the collectives): bucketed gradient
    all-reduce on the side stream + SyncBN statistics + mean over ranks must reproduce the gradient of ONE process run on
    the concatenated batch (DDP + SyncBatchNorm semantics), and the replicas stay bit-identical
    after optimizer steps"""
    import socket
    import torch.multiprocessing as mp
    over = dict(d_model=64, n_heads=4, n_layers=2, dropout=0.0, dropout_pre_encoder=0.0, dropout_att=0.0)
    vocab = 20
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    monkeypatch.setenv("MI355X_SYNCBN_MAILBOX", "1" if mailbox else "0")  # (spawned ranks inherit the environment)
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path), over, vocab), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt"); r1 = torch.load(tmp_path / "rank1.pt")
    if mailbox:
        if r0["mailbox"] is None:
            assert r1["mailbox"] is None
            pytest.skip("hipIpc handles cannot be exported / mapped on this box: the ranks stayed on the process group")
        for r in (r0, r1):  # 3 training forwards + 3 backwards x 2 layers, none timed out, every launch completed
            done, missing, kind, issued = r["mailbox"]
            assert missing == 0 and done == issued and issued >= 12, r["mailbox"]
    else:
        assert r0["mailbox"] is None
    torch.manual_seed(5)
    model = _model(over, vocab=vocab).to(dev).train()
    model.setup_optimization(dict(name="adamw", lr=1e-3, betas=[0.9, 0.98], weight_decay=0.0))
    audio, alen, tok, tl = R.synthetic_batch(4, 1.0, vocab=vocab, seed=8)
    batch = [audio.to(dev), alen.to(dev), tok.to(dev), tl.to(dev)]
    model._optimizer.zero_grad()
    loss = model.training_step(batch)["loss"]
    loss.backward()
    torch.cuda.synchronize()
    assert abs(0.5 * (r0["loss"] + r1["loss"]).item() - loss.item()) <= 1e-5 * abs(loss.item())
    for g0, g1, fp in zip(r0["grads"], r1["grads"], model.flats()):
        assert torch.equal(g0, g1)                              # the all-reduce left both ranks with the same buffer
        ref = fp.grad.detach().cpu()
        assert (g0 - ref).norm() <= 2e-4 * ref.norm(), ((g0 - ref).norm() / ref.norm())
    bn = model.encoder.layers[0].conv.batch_norm.running_mean.detach().cpu()
    assert torch.allclose(r0["bn"], bn, atol=1e-5) and torch.equal(r0["bn"], r1["bn"])  # SyncBN: global statistics
    for a, b in zip(r0["flat"], r1["flat"]):
        assert torch.equal(a, b)


def test_data_parallel_transducer_two_ranks_equal_one_process_on_the_joint_batch(tmp_path):
    """the same for EncDecRNNTModel: three flat buffers (encoder, prediction network, joint) exchanged in buckets, the
    prediction network's backward on its own stream, joint gradients produced inside the fused forward -- two ranks (2 + 2
    utterances, sub-batches of 2) against one process on the 4 utterances; replicas identical after optimizer steps"""
    import socket
    import torch.multiprocessing as mp
    over = dict(d_model=64)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path), over, 30, "rnnt"), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt"); r1 = torch.load(tmp_path / "rank1.pt")
    model = _dp_model("rnnt", over, 30)
    model.setup_optimization(dict(name="adamw", lr=1e-3, betas=[0.9, 0.98], weight_decay=0.0))
    audio, alen, tok, tl = _dp_batch("rnnt", 30)
    batch = [audio.to(dev), alen.to(dev), tok.to(dev), tl.to(dev)]
    model._optimizer.zero_grad()
    loss = model.training_step(batch)["loss"]
    loss.backward()
    model._after_backward()
    torch.cuda.synchronize()
    assert abs(0.5 * (r0["loss"] + r1["loss"]).item() - loss.item()) <= 1e-5 * abs(loss.item())
    assert len(r0["grads"]) == 3
    for g0, g1, fp in zip(r0["grads"], r1["grads"], model.flats()):
        assert torch.equal(g0, g1)
        ref = fp.grad.detach().cpu()
        assert (g0 - ref).norm() <= 3e-4 * ref.norm(), ((g0 - ref).norm() / ref.norm())
    for a, b in zip(r0["flat"], r1["flat"]):
        assert torch.equal(a, b)


def test_data_parallel_squeezeformer_two_ranks_equal_one_process_on_the_joint_batch(tmp_path):
    """and for the Squeezeformer encoder (SyncBN on the 2*d-channel statistics, time reduction / recovery, padded heads:
    d = 36, d_k = 9) under the same two-rank exchange"""
    import socket
    import torch.multiprocessing as mp
    over = dict(d_model=36, n_heads=4, n_layers=3, time_reduce_idx=1, conv_kernel_size=5, dropout=0.0, dropout_att=0.0)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path), over, 20, "squeezeformer"), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt"); r1 = torch.load(tmp_path / "rank1.pt")
    model = _dp_model("squeezeformer", over, 20)
    model.setup_optimization(dict(name="adamw", lr=1e-3, betas=[0.9, 0.98], weight_decay=0.0))
    audio, alen, tok, tl = _dp_batch("squeezeformer", 20)
    batch = [audio.to(dev), alen.to(dev), tok.to(dev), tl.to(dev)]
    model._optimizer.zero_grad()
    loss = model.training_step(batch)["loss"]
    loss.backward()
    torch.cuda.synchronize()
    assert abs(0.5 * (r0["loss"] + r1["loss"]).item() - loss.item()) <= 1e-5 * abs(loss.item())
    for g0, g1, fp in zip(r0["grads"], r1["grads"], model.flats()):
        assert torch.equal(g0, g1)
        ref = fp.grad.detach().cpu()
        assert (g0 - ref).norm() <= 3e-4 * ref.norm(), ((g0 - ref).norm() / ref.norm())
    bn = model.encoder.layers[0].conv.batch_norm.running_mean.detach().cpu()
    assert torch.allclose(r0["bn"], bn, atol=1e-5) and torch.equal(r0["bn"], r1["bn"])
    for a, b in zip(r0["flat"], r1["flat"]):
        assert torch.equal(a, b)


def test_training_wer_validation_and_transcribe(tmp_path):
    """a13's periodic training WER (ctc_models.py:591-600), validation_pass / multi_validation_epoch_end (:633-676,
    asr_model.py:95-123), predict_step (:604-631) and transcribe() on a synthetic WAV corpus"""
    import wave
    from nemo_amd.data import load_audio
    vocab = [" "] + list("abcdefghijklmnopqrs")
    m = _wav_corpus(tmp_path, 10, vocab, seed=4)
    over = dict(d_model=64, n_heads=4, n_layers=2, dropout=0.0, dropout_pre_encoder=0.0, dropout_att=0.0)
    torch.manual_seed(9)
    cfg_kw = dict(labels=vocab, log_every_n_steps=2)
    model = _model(over, vocab=len(vocab), **cfg_kw)
    model._cfg["decoder"]["vocabulary"] = vocab
    from nemo_amd.models import EncDecCTCModel
    model = EncDecCTCModel(model._cfg).to(dev).train()
    model.setup_optimization(dict(name="adamw", lr=1e-3, betas=[0.9, 0.98], weight_decay=0.0))
    dl = model.setup_training_data(dict(manifest_filepath=m, batch_size=5, shuffle=False))
    logs = []
    for i, b in enumerate(dl):
        logs.append(model.training_step([t.to(dev) for t in b], i)["log"])
    assert "training_batch_wer" not in logs[0] and "training_batch_wer" in logs[1]
    assert 0.0 <= logs[1]["training_batch_wer"] < 50.0
    # validation: loss mean over batches, WER = total edit distance / total reference words
    model.setup_validation_data(dict(manifest_filepath=m, batch_size=4))
    res = model.validate()
    outs = model.validation_step_outputs
    assert len(outs) == 3 and model.training
    assert abs(res["val_loss"].item() - float(np.mean([o["val_loss"].item() for o in outs]))) < 1e-4
    assert abs(res["val_wer"] - sum(o["val_wer_num"] for o in outs) / sum(o["val_wer_denom"] for o in outs)) < 1e-12
    # transcribe: paths and waveforms give the same texts, in input order, equal to decode(forward(padded batch))
    paths = [str(tmp_path / f"u{i}.wav") for i in (3, 0, 7, 5, 1)]
    texts = model.transcribe(paths, batch_size=5)
    waves = [load_audio(p_, 16000) for p_ in paths]
    assert model.transcribe([w.numpy() for w in waves], batch_size=5) == texts and model.training
    hyp = model.transcribe(paths[:2], batch_size=5, return_hypotheses=True)
    assert [h[0] for h in hyp] == model.transcribe(paths[:2], batch_size=5) and all(isinstance(h[1], list) for h in hyp)
    model.eval()
    feat = model.preprocessor.featurizer
    d0, feat.dither = feat.dither, 0.0
    lens = torch.tensor([w.numel() for w in waves])
    sig = torch.zeros(5, int(lens.max()))
    for r, w in enumerate(waves):
        sig[r, : w.numel()] = w
    lp, el, _ = model.forward(input_signal=sig.to(dev), input_signal_length=lens.to(dev))
    feat.dither = d0
    assert model.wer.decoding(lp, el) == texts
    # predict_step keeps the sample ids next to the texts
    ds = dl.dataset
    ds.return_sample_id = True
    batch = ds._collate_fn([ds[i] for i in (2, 4)])
    pred = model.predict_step([t.to(dev) if torch.is_tensor(t) else t for t in batch])
    assert [int(i) for i, _ in pred] == [2, 4] and all(isinstance(t, str) for _, t in pred)


def _ragged_bn_worker(rank, world, port, out_dir, over, vocab, secs):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        torch.manual_seed(5)
        model = _model(over, vocab=vocab).to(dev).train()
        audio, alen, tok, tl = R.synthetic_batch(2, secs[rank], vocab=vocab, seed=40 + rank)
        loss = model.training_step([audio.to(dev), alen.to(dev), tok.to(dev), tl.to(dev)])["loss"]
        loss.backward()
        torch.cuda.synchronize()
        bn = model.encoder.layers[0].conv.batch_norm
        torch.save(dict(mean=bn.running_mean.detach().cpu(), var=bn.running_var.detach().cpu(),
                        grad=model.encoder.layers[0].conv.depthwise_conv.weight.grad.detach().cpu()),
                   os.path.join(out_dir, f"ragged{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_syncbn_counts_are_exact_for_ranks_with_different_padded_lengths(tmp_path):
    """torch.nn.SyncBatchNorm divides the summed statistics by the SUM of the ranks' element counts.  Two ranks holding
    batches padded to different lengths (1.0 s vs 0.6 s: T' = 26 vs 16): the first layer's synchronised mean / variance
    must equal the count-weighted combination of the two ranks' own statistics (each measured in a one-process run)."""
    import socket
    import torch.multiprocessing as mp
    over = dict(d_model=64, n_heads=4, n_layers=1, dropout=0.0, dropout_pre_encoder=0.0, dropout_att=0.0)
    vocab, secs = 20, (1.0, 0.6)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_ragged_bn_worker, args=(2, port, str(tmp_path), over, vocab, secs), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "ragged0.pt"); r1 = torch.load(tmp_path / "ragged1.pt")
    assert torch.equal(r0["mean"], r1["mean"]) and torch.equal(r0["var"], r1["var"])
    stats = []
    for rank in (0, 1):  # the ranks' own statistics: one process, no group
        torch.manual_seed(5)
        model = _model(over, vocab=vocab).to(dev).train()
        audio, alen, tok, tl = R.synthetic_batch(2, secs[rank], vocab=vocab, seed=40 + rank)
        model.training_step([audio.to(dev), alen.to(dev), tok.to(dev), tl.to(dev)])
        torch.cuda.synchronize()
        bn = model.encoder.layers[0].conv.batch_norm
        n = 2 * (((audio.shape[1] // 160 + 1) - 1) // 2 // 2 + 1)  # B x T' (tensor length after the two stride-2 convs)
        mean = bn.running_mean.double().cpu() / 0.1
        var_unb = (bn.running_var.double().cpu() - 0.9) / 0.1
        stats.append((n, mean, var_unb * (n - 1) / n + mean ** 2))  # (count, E[x], E[x^2])
    (na, ma, qa), (nb, mb, qb) = stats
    assert na != nb
    n = na + nb
    mean = (na * ma + nb * mb) / n
    var = (na * qa + nb * qb) / n - mean ** 2
    assert torch.allclose(r0["mean"].double(), 0.1 * mean, atol=1e-6, rtol=1e-5)
    assert torch.allclose(r0["var"].double(), 0.9 + 0.1 * var * n / (n - 1), atol=1e-6, rtol=1e-5)
    # with the old "count x world" rule rank 0 would have divided by 2*na and rank 1 by 2*nb
    wrong = (na * ma + nb * mb) / (2 * na)
    assert not torch.allclose(r0["mean"].double(), 0.1 * wrong, atol=1e-6, rtol=1e-4)


def test_frozen_parameters_are_neither_updated_nor_decayed():
    """requires_grad=False / freeze() must keep a parameter out of the fused AdamW (torch skips grad-less
    parameters): a frozen tensor keeps its bits -- no update, no decoupled weight decay -- while the rest trains"""
    over = dict(d_model=64, n_heads=4, n_layers=2, dropout=0.0, dropout_pre_encoder=0.0, dropout_att=0.0)
    torch.manual_seed(1)
    model = _model(over, vocab=20).to(dev).train()
    model.setup_optimization(dict(name="adamw", lr=1e-2, betas=[0.9, 0.98], weight_decay=0.1))
    frozen = [model.encoder.layers[0].feed_forward1.linear1.weight, model.encoder.layers[1].norm_out.bias,
              model.decoder.decoder_layers[0].bias]
    for p in frozen:
        p.requires_grad_(False)
    model.flats()
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    audio, alen, tok, tl = R.synthetic_batch(2, 1.0, vocab=20, seed=8)
    batch = [audio.to(dev), alen.to(dev), tok.to(dev), tl.to(dev)]
    for _ in range(3):
        model.fit_step(batch)
    torch.cuda.synchronize()
    ids = {id(p) for p in frozen}
    moved = 0
    for n, p in model.named_parameters():
        if id(p) in ids:
            assert torch.equal(p.detach(), before[n]), n
        else:
            moved += int(not torch.equal(p.detach(), before[n]))
    assert moved > 50


def _encoder_vs_fixture(z, enc, rtol_y=2e-4, rtol_g=2e-3):
    """drop-in encoder (train mode = batch-statistics BatchNorm, dropout 0) against a fixture made by the reference's own class:
    output, lengths and every parameter gradient of the fixed linear functional sum(y * w * valid)"""
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("P.")}
    missing, unexpected = enc.load_state_dict(sd, strict=False)
    assert not unexpected and not [m for m in missing if "pos_enc" not in m], (missing, unexpected)
    enc = enc.to(dev).train()
    enc.flat_parameters().zero_grad()
    x = torch.from_numpy(z["x"]).to(dev)
    length = torch.from_numpy(z["length"]).to(dev)
    y, yl = enc(audio_signal=x, length=length)
    assert np.array_equal(yl.cpu().numpy(), z["y_len"])
    yr = torch.from_numpy(z["y"])
    valid = (torch.arange(yr.shape[2]).unsqueeze(0) < torch.from_numpy(z["y_len"]).unsqueeze(1)).unsqueeze(1)
    err = ((y.detach().float().cpu() - yr) * valid).abs().max().item() / yr.abs().max().item()
    assert err < rtol_y, err
    w = torch.from_numpy(z["w"])
    (y * (w * valid).to(dev)).sum().backward()
    torch.cuda.synchronize()
    got = {n: p.grad.detach().float().cpu() for n, p in enc.named_parameters()}
    ref = {k[2:]: z[k] for k in z.files if k.startswith("G.")}
    assert set(ref) == set(got), set(ref) ^ set(got)
    _cmp_grads(got, ref, rtol=rtol_g, floor=1e-3)


def test_fastconformer_encoder_dw_striding_x8_matches_reference_fixture(golden_dir):
    """BASELINE.json configs[3]'s encoder geometry (FastConformer: 'dw_striding' x8 sub-sampling with its own channel count,
    depthwise kernel 9) against tests/golden/ref_fastconformer_tiny.npz (the reference ConformerEncoder run through the
    shim): depthwise stride-2 conv kernels forward / data-gradient / weight-gradient, pointwise GEMMs with ReLU + mask."""
    from nemo_amd.modules import ConformerEncoder
    z = np.load(os.path.join(golden_dir, "ref_fastconformer_tiny.npz"))
    enc = ConformerEncoder(feat_in=40, n_layers=2, d_model=32, feat_out=-1, subsampling="dw_striding", subsampling_factor=8,
                           subsampling_conv_channels=16, ff_expansion_factor=4, self_attention_model="rel_pos", n_heads=4,
                           conv_kernel_size=9, dropout=0.0, dropout_pre_encoder=0.0, dropout_emb=0.0, dropout_att=0.0)
    _encoder_vs_fixture(z, enc)


def test_fastconformer_geometry_bf16_runs_the_production_paths():
    """d = 256 / 4 heads (d_k = 64: fused flash attention), 256 sub-sampling channels, x8, kernel 9, bf16: finite loss and
    gradients, and the loss of the bf16 path close to the fp32 path of the same model (same weights, same batch)"""
    from nemo_amd.modules import ConformerEncoder
    kw = dict(feat_in=80, n_layers=2, d_model=256, subsampling="dw_striding", subsampling_factor=8,
              subsampling_conv_channels=256, n_heads=4, conv_kernel_size=9, dropout=0.0, dropout_pre_encoder=0.0,
              dropout_emb=0.0, dropout_att=0.0)
    torch.manual_seed(4)
    e32 = ConformerEncoder(compute_dtype=torch.float32, **kw)
    e16 = ConformerEncoder(compute_dtype=torch.bfloat16, **kw)
    e16.load_state_dict(e32.state_dict())
    e32, e16 = e32.to(dev).train(), e16.to(dev).train()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 80, 640, generator=g).to(dev)
    length = torch.tensor([640, 500, 333]).to(dev)
    outs = []
    for e in (e32, e16):
        e.flat_parameters().zero_grad()
        y, yl = e(audio_signal=x, length=length)
        assert y.shape == (3, 256, 80) and yl.tolist() == [80, 63, 42]
        (y.float() ** 2).mean().backward()
        torch.cuda.synchronize()
        outs.append((y.detach().float(), e.flat_parameters().grad.detach().clone()))
        assert torch.isfinite(outs[-1][1]).all()
    (y32, g32), (y16, g16) = outs
    assert (y16 - y32).norm() / y32.norm() < 3e-2
    assert torch.dot(g16, g32) / (g16.norm() * g32.norm()) > 0.99


def _transducer_head(z, fused, dtype=None):
    from nemo_amd.modules import RNNTDecoder, RNNTJoint, RNNTLoss
    V, H, D, J = 12, 16, 24, 20
    dec = RNNTDecoder(prednet={"pred_hidden": H, "pred_rnn_layers": 2, "dropout": 0.0}, vocab_size=V, compute_dtype=dtype)
    kw = dict(fuse_loss_wer=True, fused_batch_size=2) if fused else {}
    joint = RNNTJoint(jointnet={"encoder_hidden": D, "pred_hidden": H, "joint_hidden": J, "activation": "relu", "dropout": 0.0},
                      num_classes=V, compute_dtype=dtype, **kw)
    dec.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("P.D.")})
    joint.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("P.J.")})
    loss = RNNTLoss(blank=V, reduction="sum")
    if fused:
        joint.set_loss(loss)
    return dec.to(dev).train(), joint.to(dev).train(), loss


@pytest.mark.parametrize("fused", [False, True])
def test_transducer_head_matches_reference_fixture(golden_dir, fused):
    """RNNTDecoder (embedding + SOS frame + 2-layer LSTM) + RNNTJoint + RNN-T loss on a ragged batch (one empty label
    sequence) against tests/golden/ref_transducer_tiny.npz, produced by the reference's RNNTDecoder / RNNTJoint /
    RNNTLossPytorch: decoder output, logits, loss, gradient w.r.t. the encoder output and every parameter.  `fused` = the
    recipe's training path (joint.fuse_loss_wer with sub-batches of 2: the logits never exist for the whole batch)."""
    z = np.load(os.path.join(golden_dir, "ref_transducer_tiny.npz"))
    dec, joint, loss_mod = _transducer_head(z, fused)
    V = 12
    enc = torch.from_numpy(z["enc"]).to(dev).requires_grad_(True)
    enc_len = torch.from_numpy(z["enc_len"]).to(dev)
    tgt = torch.from_numpy(z["targets"]).to(dev)
    tgt_len = torch.from_numpy(z["tgt_len"]).to(dev)
    for m in (dec, joint):
        m.flat_parameters().zero_grad()
    g, _, _ = dec(targets=tgt, target_length=tgt_len)
    assert np.abs(g.detach().cpu().numpy() - z["dec_out"]).max() < 2e-5
    if fused:
        loss, _, _, _ = joint(encoder_outputs=enc, decoder_outputs=g, encoder_lengths=enc_len, transcripts=tgt,
                              transcript_lengths=tgt_len)
    else:
        logits = joint(encoder_outputs=enc, decoder_outputs=g)
        # (the fixture was produced on the CPU, where the reference joint applies log_softmax itself, rnnt.py:1700-1712; on
        #  the GPU -- and here -- the joint returns logits and the loss fuses the normalisation)
        assert np.abs(torch.log_softmax(logits.detach(), -1).cpu().numpy() - z["logits"]).max() < 5e-5
        loss = loss_mod(logits, tgt.clamp(max=V - 1), enc_len, tgt_len)
    assert abs(float(loss) - float(z["loss"])) <= 1e-4 * abs(float(z["loss"])), (float(loss), float(z["loss"]))
    loss.sum().backward()
    torch.cuda.synchronize()
    assert np.abs(enc.grad.cpu().numpy() - z["d_enc"]).max() <= 2e-4 * np.abs(z["d_enc"]).max() + 1e-6
    got = {"D." + n: p.grad.detach().float().cpu() for n, p in dec.named_parameters()}
    got.update({"J." + n: p.grad.detach().float().cpu() for n, p in joint.named_parameters()})
    ref = {k[2:]: z[k] for k in z.files if k.startswith("G.")}
    assert set(ref) == set(got), set(ref) ^ set(got)
    for k, r in ref.items():
        s = max(np.abs(r).max(), 1e-4)
        assert np.abs(got[k].numpy() - r).max() <= 1e-3 * s, (k, np.abs(got[k].numpy() - r).max(), s)
    # the padding row of the embedding receives no gradient (torch.nn.Embedding(padding_idx))
    assert float(got["D.prediction.embed.weight"][V].abs().max()) == 0.0


@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
def test_fused_joint_cuts_sub_batches_to_their_own_lengths(cdt):
    """rnnt.py:1559-1600: the fused joint + loss cuts every sub-batch to its longest encoder / target length.  On a batch that is
    padded well beyond its utterances (audio AND transcripts) the cut path gives the loss and every gradient of the padded
    grid (cells beyond an utterance's lengths carry neither), fp32 to round-off and bf16 to its own tolerance; the un-fused
    path accepts the over-padded transcript too (losses/rnnt.py:446-484 narrowing)."""
    audio, alen, tok, tl = R.synthetic_batch(4, 2.0, vocab=30, seed=12)
    alen = torch.tensor([20000, 16000, 30000, 12000]); tl = torch.tensor([3, 2, 5, 1])
    tok = torch.cat([tok, torch.zeros(4, 3, dtype=tok.dtype)], 1)  # transcripts padded 3 beyond the longest
    batch = [audio.to(dev), alen.to(dev), tok.to(dev), tl.to(dev)]
    torch.manual_seed(2)
    m = _rnnt_model(cdt, d_model=64).to(dev).train()
    if cdt == torch.bfloat16:
        m.decoder.compute_dtype = m.joint.compute_dtype = torch.bfloat16
    res = {}
    for cut in (False, True):
        m.joint.truncate_sub_batches = cut
        for mod in (m.encoder, m.decoder, m.joint):
            mod.flat_parameters().zero_grad()
        loss = m.training_step(batch)["loss"]
        loss.backward()
        torch.cuda.synchronize()
        res[cut] = (loss.item(), {n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None})
    tol = 2e-4 if cdt == torch.float32 else 2e-2  # (fp32: the cut changes the GEMM shapes, i.e. split-K and atomic summation order)
    assert abs(res[True][0] - res[False][0]) <= tol * abs(res[False][0]), (res[True][0], res[False][0])
    scale = max(g.norm().item() for g in res[False][1].values())
    for n, g in res[False][1].items():
        err = (res[True][1][n] - g).norm().item()
        assert err <= tol * max(g.norm().item(), 1e-2 * scale) * (1 if cdt == torch.float32 else 3), (n, err, g.norm().item())
    if cdt == torch.float32:
        m.joint.set_fuse_loss_wer(False)
        lu = m.training_step(batch)["loss"]
        torch.cuda.synchronize()
        assert abs(lu.item() - res[False][0]) <= 1e-4 * abs(res[False][0]), (lu.item(), res[False][0])


def _rnnt_model(cdt=None, **enc_over):
    from nemo_amd.models import EncDecRNNTModel, fastconformer_transducer_config
    over = dict(d_model=64, n_heads=4, n_layers=2, subsampling_conv_channels=32, dropout=0.0, dropout_pre_encoder=0.0,
                dropout_att=0.0)
    over.update(enc_over)
    if cdt is not None:
        over["compute_dtype"] = cdt
    cfg = fastconformer_transducer_config("small", vocab_size=30, **over)
    cfg["preprocessor"]["dither"] = 0.0
    cfg["decoder"]["prednet"].update(pred_hidden=64, dropout=0.0)
    cfg["joint"]["jointnet"].update(joint_hidden=64, dropout=0.0)
    cfg["joint"]["fused_batch_size"] = 2
    m = EncDecRNNTModel(cfg)
    if cdt is not None:
        m.decoder.compute_dtype = m.joint.compute_dtype = cdt
    return m


def test_fastconformer_transducer_model_trains_and_bf16_tracks_fp32():
    """EncDecRNNTModel end to end (BASELINE.json configs[3] in miniature): x8 dw_striding encoder -> LSTM prediction network
    -> fused joint + RNN-T loss in sub-batches of 2 -> backward through all three -> fused AdamW.  fp32: the loss falls;
    bf16 (d_k = 64: flash attention, MFMA GEMMs everywhere incl. the joint) starts from the same loss within 1 %."""
    audio, alen, tok, tl = R.synthetic_batch(4, 2.0, vocab=30, seed=12)
    alen = torch.tensor([32000, 28000, 30000, 20000]); tl = torch.tensor([6, 4, 5, 3])
    batch = [audio.to(dev), alen.to(dev), tok.to(dev), tl.to(dev)]
    torch.manual_seed(2)
    m32 = _rnnt_model(torch.float32, d_model=256).to(dev).train()
    m32.setup_optimization(dict(name="adamw", lr=1e-3, betas=[0.9, 0.98], weight_decay=0.0))
    sd = {k: v.clone() for k, v in m32.state_dict().items()}
    losses = [m32.fit_step(batch)["loss"].item() for _ in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < 0.9 * losses[0], losses
    m16 = _rnnt_model(torch.bfloat16, d_model=256)
    m16.load_state_dict(sd)
    m16 = m16.to(dev).train()
    l16 = m16.training_step(batch)["loss"]
    l16.backward()
    torch.cuda.synchronize()
    assert abs(l16.item() - losses[0]) <= 1e-2 * abs(losses[0]), (l16.item(), losses[0])
    for n, p in m16.named_parameters():
        assert torch.isfinite(p.grad).all(), n
    # validation: the fused joint computes the loss only (no backward GEMMs) -- same value as the training path in eval mode
    m16.eval()
    v = m16.validation_pass(batch)["val_loss"]
    e = m16.training_step(batch)["loss"]
    torch.cuda.synchronize()
    assert torch.isfinite(v) and abs(v.item() - e.item()) <= 1e-5 * abs(e.item()), (v.item(), e.item())
    m16.train()
    # dropout on (recipe values): runs, finite, stochastic
    torch.manual_seed(3)
    md = _rnnt_model(torch.bfloat16, d_model=256, dropout=0.1, dropout_att=0.1)
    md._cfg["decoder"]["prednet"]["dropout"] = 0.2
    from nemo_amd.models import EncDecRNNTModel
    cfg = md._cfg; cfg["joint"]["jointnet"]["dropout"] = 0.2
    md = EncDecRNNTModel(cfg)
    md.decoder.compute_dtype = md.joint.compute_dtype = torch.bfloat16
    md = md.to(dev).train()
    a, b = md.training_step(batch)["loss"], md.training_step(batch)["loss"]
    torch.cuda.synchronize()
    assert torch.isfinite(a) and torch.isfinite(b) and a.item() != b.item()


@pytest.mark.parametrize("mailbox", ["0", "1"])
def test_bench_eight_rank_rehearsal_on_one_gpu(mailbox):
    """the driver's 8-GPU command line, rehearsed on ONE GPU (BENCH_DEVICE=0: all ranks on device 0, gloo carries the collectives;
    VERDICT r5 item 7): `python bench.py --gpus 8` self-launches eight ranks through torch.distributed.run, every rank steps the
    small recipe with SyncBatchNorm (process group, or the peer-mapped mailboxes with MI355X_SYNCBN_MAILBOX=1), the bucketed gradient
    exchange with its tail bucket, the max-over-ranks timing -- and rank 0 prints exactly one JSON line for n_gpus = 8"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(BENCH_DEVICE="0", BENCH_DIST_BACKEND="gloo", MI355X_SYNCBN_MAILBOX=mailbox, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--size", "small", "--batch", "2", "--secs", "2",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["steps"] == 2 and j["value"] > 0 and j["scaling"] == "weak"
    d = j["distributed"]
    assert d["world_size"] == 8 and len(d["ranks_seen"]) == 8 and sorted(x["rank"] for x in d["ranks_seen"]) == list(range(8))
    assert d["valid"] is False and d["rehearsal"] is True   # ranks share one GPU: never a scaling number
    assert d["syncbn_exchange"] == ("mailbox" if mailbox == "1" else "process group") or mailbox == "1"   # (mailbox: all-or-nothing, may decline)
    assert j["config"]["global_batch"] == 16
