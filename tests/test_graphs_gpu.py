"""-m gpu: the encoder's forward / backward launch sequence recorded as hipGraph segments (nemo_amd/graphs.py) against the SAME
sequence issued launch by launch from Python (the eager sequencer, which the parity tests pin to the oracle).  A replayed step
must be the eager step: same kernels, same arguments, same order -- only float-atomic order inside the split-K weight gradients
may differ.  Reference analogue: whole-step CUDA-graph capture, nemo/utils/callbacks/cuda_graph.py:251."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import conformer_ref as R

dev = "cuda"
NODROP = dict(dropout=0.0, dropout_pre_encoder=0.0, dropout_att=0.0, dropout_emb=0.0)


def _model(over, vocab=20, size="small", spec_augment=False, **kw):
    from nemo_amd.models import EncDecCTCModel, conformer_ctc_config
    cfg = conformer_ctc_config(size, vocab_size=vocab, spec_augment=spec_augment, **over)
    cfg["preprocessor"]["dither"] = 0.0
    cfg.update(kw)
    return EncDecCTCModel(cfg)


def _batch(B=4, secs=1.0, vocab=20, seed=8, lens=None):
    audio, alen, tok, tl = R.synthetic_batch(B, secs, vocab=vocab, seed=seed)
    if lens is not None:
        alen = torch.tensor(lens)
    return [audio.to(dev), alen.to(dev), tok.to(dev), tl.to(dev)]


REPLAYS = ["tape", "graph"]   # a recorded segment runs again as a launch tape (csrc/tape.hip, the default) or as a hipGraph replay


def _check_replay_kind(info, replay):
    """the recording really runs the way the test asked for; a tape re-issues every node itself (no segment fell back to a graph)"""
    for rec in info:
        assert rec["replay"] == ("launch tape" if replay == "tape" else "hipGraph"), rec
        if replay == "tape":
            live_bwd = os.environ.get("MI355X_GRAPHS_BWD_LIVE", "0") != "0"   # (forward-only replay: the backward has no recording)
            for t in (rec["fwd_tape"],) + (() if live_bwd else (rec["bwd_tape"],)):
                assert t is not None and t["graph_fallbacks"] == 0 and t["kernels"] > 0 and t["lanes"] >= 1, rec
        else:
            assert rec["fwd_tape"] is None and rec["bwd_tape"] is None, rec


def _run(over, graphs, steps, batches, dtype=torch.float32, lr=1e-3, seed=21, hook=False, replay="tape"):
    torch.manual_seed(seed)
    model = _model(dict(over, compute_dtype=dtype)).to(dev).train()
    model.decoder.compute_dtype = dtype
    model.encoder.use_graphs = graphs
    model.encoder.graph_tape = replay == "tape"
    model.encoder.graph_auto = False  # forced: record after `graph_warmup` eager steps (the default would time both ways first)
    model.optimizer_in_backward = hook
    model.setup_optimization(dict(name="adamw", lr=lr, betas=[0.9, 0.98], weight_decay=1e-3))
    losses = [model.fit_step(batches[i % len(batches)])["loss"].item() for i in range(steps)]
    torch.cuda.synchronize()
    return model, losses


def _same_trajectory(l_a, l_b, base=5e-5, growth=0.25):
    """two runs of the same training steps that differ only in HOW the launches were issued.  The weight gradients accumulate
    through fp32 atomics (split-K slices, tap / bias partials): their order, hence the last bits, depend on the timing, and a
    trajectory amplifies that from step to step (seen: 6.4e-5 relative at step 30 of 32, one run in three) -- so the tolerance starts
    at `base` and grows by `growth` x base per step.  A replay that reads a wrong arena address is off by orders of magnitude."""
    for i, (a, b) in enumerate(zip(l_a, l_b)):
        assert abs(a - b) <= base * (1.0 + growth * i) * abs(b), (i, l_a, l_b)


@pytest.mark.parametrize("replay", REPLAYS)
def test_recorded_sequence_is_the_eager_sequence_fp32(replay):
    over = dict(d_model=64, n_heads=4, n_layers=3, **NODROP)
    batches = [_batch(lens=[16000, 12000, 16000, 9000])]
    m_g, l_g = _run(over, True, 7, batches, replay=replay)
    m_e, l_e = _run(over, False, 7, batches)
    info = m_g.encoder.graph_info()
    _check_replay_kind(info, replay)
    assert len(info) == 1 and info[0]["fwd_graphs"] == 1 and info[0]["bwd_graphs"] == 1 and info[0]["bwd_host_calls"] == 0, info
    assert m_e.encoder.graph_info() == []
    # two runs differ from their FIRST update on (float-atomic order inside the split-K weight gradients: the two eager warm-up
    # steps already disagree in the last bit) and AdamW's first steps amplify that: the bound grows with the step
    for i, (a, b) in enumerate(zip(l_g, l_e)):
        assert abs(a - b) <= 2e-5 * (1 + 2 * i) * abs(b), (l_g, l_e)
    # (the weights themselves: AdamW turns the float-atomic noise of analytically-zero gradients -- key bias, depthwise bias
    # under batch statistics -- into full +-lr steps of random sign, in the eager run as much as in the replayed one, so two
    # runs agree on them only to a fraction of one step, 7 x lr x sqrt(count of such elements); measured 8.6e-4)
    for fa, fb in zip(m_g.flats(), m_e.flats()):
        assert (fa.flat - fb.flat).norm() <= 3e-3 * fb.flat.norm()
    # BatchNorm bookkeeping happens inside the recorded forward too
    assert int(m_g.encoder.layers[0].conv.batch_norm.num_batches_tracked) == 7
    assert torch.allclose(m_g.encoder.layers[0].conv.batch_norm.running_var, m_e.encoder.layers[0].conv.batch_norm.running_var, rtol=1e-5)


@pytest.mark.parametrize("replay", REPLAYS)
def test_recorded_sequence_with_zero_padded_heads_bf16(replay):
    """d_k = 24 runs the fused attention through heads zero-padded to 64 lanes: the linear_pos weight gradients are then per-layer
    batched GEMMs issued from INSIDE the side-stream scope of the positional gradients.  A nested scope used to make the side stream
    wait for its own event -- harmless live, but inside a stream capture hip::Stream::EndCapture recursed until the stack ended
    (`bench.py --size small` died with SIGSEGV under MI355X_GRAPHS=auto / 1).  Recorded and replayed here, against live launches."""
    over = dict(d_model=48, n_heads=2, n_layers=2, **NODROP)
    batches = [_batch(lens=[16000, 12000, 16000, 9000])]
    m_g, l_g = _run(over, True, 6, batches, dtype=torch.bfloat16, replay=replay)
    m_e, l_e = _run(over, False, 6, batches, dtype=torch.bfloat16)
    assert m_g.encoder._geometry(torch.bfloat16)[1] == 64 and m_g.encoder.d_k == 24
    info = m_g.encoder.graph_info()
    _check_replay_kind(info, replay)
    assert len(info) == 1 and info[0]["bwd_graphs"] >= 1, info
    for i, (a, b) in enumerate(zip(l_g, l_e)):
        assert abs(a - b) <= 2e-3 * (1 + i) * abs(b), (l_g, l_e)


@pytest.mark.parametrize("replay", REPLAYS)
def test_recorded_sequence_with_hooks_between_the_segments(replay):
    """optimizer-behind-backward installs a per-layer hook: the backward sequence is cut at every hook, which stays a live call"""
    over = dict(d_model=64, n_heads=4, n_layers=3, **NODROP)
    batches = [_batch()]
    m_g, l_g = _run(over, True, 7, batches, hook=True, replay=replay)
    m_e, l_e = _run(over, False, 7, batches, hook=True)
    info = m_g.encoder.graph_info()
    _check_replay_kind(info, replay)
    assert info and info[0]["bwd_host_calls"] >= 3 + 2 and info[0]["bwd_graphs"] == info[0]["bwd_host_calls"] + 1, info
    for a, b in zip(l_g, l_e):
        assert abs(a - b) <= 1e-3 * abs(b), (l_g, l_e)
    assert l_g[-1] < 0.95 * l_g[0]


def test_tape_replay_orders_the_linear_pos_gradient_behind_a_slow_weight_gradient_lane():
    """with the per-layer hook (optimizer behind backward) a tape is replayed with join_between=False; the
    post-loop join in front of the linear_pos weight gradient was a no-op in the recording (the last cut had cleared the fork
    flag), so the main lane could read dp_all while the side lane (the dpos kernels of the fused d_k = 64 path) was still
    running.  Here the side stream is held back by a long sleep in front of every replayed step and two batches alternate, so a
    main lane that runs ahead reads the OTHER batch's dp_all.  lr = 0: the weights never move and Adam's first moment of
    linear_pos is a pure function of the per-step gradients."""
    over = dict(d_model=128, n_heads=2, n_layers=2, **NODROP)
    batches = [_batch(B=2, secs=1.0, seed=8), _batch(B=2, secs=1.0, seed=9)]

    def moments(graphs):
        torch.manual_seed(5)
        model = _model(dict(over, compute_dtype=torch.bfloat16)).to(dev).train()
        model.decoder.compute_dtype = torch.bfloat16
        enc = model.encoder
        enc.use_graphs, enc.graph_tape, enc.graph_auto = graphs, True, False
        model.optimizer_in_backward = True
        model.setup_optimization(dict(name="adamw", lr=0.0, betas=[0.9, 0.98], weight_decay=0.0))
        for it in range(enc.graph_warmup + 7):
            if graphs and enc._wg_stream is not None and it > enc.graph_warmup + 1:
                with torch.cuda.stream(enc._wg_stream):
                    torch.cuda._sleep(int(40e6))  # ~20 ms: far longer than the whole tiny backward
            model.fit_step(batches[it % 2])
        torch.cuda.synchronize()
        assert enc.grad_ready_hook is not None and not enc._wgrad_join_per_layer
        if graphs:
            info = enc.graph_info()
            assert info and info[0]["replay"] == "launch tape" and info[0]["bwd_tape"]["lanes"] >= 2, info
        out = {}
        for fp in model.flats():
            m, _ = model._optimizer._moments(fp)
            for n in fp.order:
                if "linear_pos" in n:
                    off, num = fp.offsets[n]
                    out[n] = m[off: off + num].detach().clone()
        return out

    m_t, m_e = moments(True), moments(False)
    assert m_e and set(m_t) == set(m_e)
    for n in m_e:
        assert float(m_e[n].norm()) > 0
        assert (m_t[n] - m_e[n]).norm() <= 2e-2 * m_e[n].norm(), n


@pytest.mark.parametrize("replay", REPLAYS)
def test_each_batch_shape_gets_its_own_recording(replay):
    over = dict(d_model=64, n_heads=4, n_layers=2, **NODROP)
    batches = [_batch(B=4, secs=1.0), _batch(B=2, secs=1.5, seed=9)]
    m_g, l_g = _run(over, True, 10, batches, replay=replay)
    m_e, l_e = _run(over, False, 10, batches)
    assert len(m_g.encoder.graph_info()) == 2
    _check_replay_kind(m_g.encoder.graph_info(), replay)
    _same_trajectory(l_g, l_e)


def _step_grads(model, batch):
    model._optimizer.zero_grad()
    loss = model.training_step(batch)["loss"]
    loss.backward()
    torch.cuda.synchronize()
    return loss.item(), [fp.grad.detach().clone() for fp in model.flats()]


@pytest.mark.parametrize("replay", REPLAYS)
def test_dropout_masks_follow_the_device_step_word_bf16(replay):
    """bf16 production kernels (fused attention, MFMA GEMM epilogues, LayerNorm-backward casts) with every dropout site on:
    (1) with the step word forced to 0 a replayed step regenerates exactly the masks of the eager step with the recorded seed, in
    forward AND backward (gradients agree); (2) consecutive replays draw different masks."""
    from nemo_amd import ops
    over = dict(d_model=64, n_heads=1, n_layers=2, dropout=0.1, dropout_pre_encoder=0.1, dropout_att=0.1, dropout_emb=0.1,
                compute_dtype=torch.bfloat16)
    batch = _batch(B=4, secs=2.0)
    torch.manual_seed(3)
    mg = _model(over).to(dev).train()
    mg.decoder.compute_dtype = torch.bfloat16
    mg.setup_optimization(dict(name="adamw", lr=0.0))
    enc = mg.encoder
    enc.graph_auto = False
    enc.graph_tape = replay == "tape"
    for _ in range(enc.graph_warmup):
        _step_grads(mg, batch)
    l3, _ = _step_grads(mg, batch)   # recorded + replayed
    assert enc.graph_info() and enc.graph_info()[0]["bwd_graphs"] == 1
    _check_replay_kind(enc.graph_info(), replay)
    if replay == "tape":  # the bf16 path sends its weight gradients to the side stream: the tape must bring that lane back
        assert enc.graph_info()[0]["bwd_tape"]["lanes"] == 2 and enc.graph_info()[0]["bwd_tape"]["events"] > 0, enc.graph_info()
    seed_rec = enc._step_seed        # the seed baked into the recorded keys
    l4, _ = _step_grads(mg, batch)
    assert l3 != l4, "two replays drew the same dropout masks"
    enc._step_word.fill_(-ops.STEP_WORD_INC)      # the forward graph adds the increment first: the kernels then read 0
    lz, gz = _step_grads(mg, batch)
    assert int(enc._step_word) == 0
    # the eager step with the same seed
    enc.use_graphs = False
    enc._step_seed = seed_rec - 1
    le, ge = _step_grads(mg, batch)
    assert abs(lz - le) <= 1e-5 * abs(le), (lz, le)
    for a, b in zip(gz, ge):
        assert (a - b).norm() <= 2e-3 * b.norm(), ((a - b).norm() / b.norm())
    assert abs(l3 - le) > 1e-6 * abs(le)          # (and a non-zero word gave other masks)


def test_backward_of_an_overwritten_forward_is_refused():
    over = dict(d_model=64, n_heads=4, n_layers=2, **NODROP)
    model = _model(over).to(dev).train()
    model.encoder.graph_auto = False
    batch = _batch()
    for _ in range(model.encoder.graph_warmup + 1):
        model.training_step(batch)["loss"].backward()
    first = model.training_step(batch)["loss"]
    model.training_step(batch)["loss"].backward()   # a later forward of the same shape owns the activations now
    with pytest.raises(RuntimeError, match="overwritten by a later forward"):
        first.backward()


def _dp_worker(rank, world, port, out_dir, over, graphs, wire="fp32"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MI355X_GRAD_WIRE"] = wire
    dist.init_process_group("gloo", rank=rank, world_size=world)  # both ranks on the one GPU; gloo stages through the host
    try:
        torch.cuda.set_device(0)
        torch.manual_seed(5)
        model = _model(over).to(dev).train()
        model.encoder.use_graphs = graphs
        model.encoder.graph_auto = False
        model.setup_optimization(dict(name="adamw", lr=1e-3, betas=[0.9, 0.98], weight_decay=0.0))
        audio, alen, tok, tl = R.synthetic_batch(4, 1.0, vocab=20, seed=8)
        alen = torch.tensor([16000, 9000, 14000, 16000])
        sl = slice(2 * rank, 2 * rank + 2)
        batch = [audio[sl].to(dev), alen[sl].to(dev), tok[sl].to(dev), tl[sl].to(dev)]
        losses = [model.fit_step(batch)["loss"].item() for _ in range(6)]
        torch.cuda.synchronize()
        torch.save(dict(losses=losses, flat=[fp.flat.detach().cpu() for fp in model.flats()], info=model.encoder.graph_info(),
                        bn=model.encoder.layers[1].conv.batch_norm.running_var.detach().cpu()),
                   os.path.join(out_dir, f"g{int(graphs)}{'' if wire == 'fp32' else wire}_rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_ranks_replay_with_live_collectives_between_the_segments(tmp_path):
    """data parallel: SyncBatchNorm all-reduces (forward and backward of every layer) and the bucketed gradient exchange stay live
    host calls between graph segments; the replicas must follow the eager two-rank run and stay identical to each other"""
    import torch.multiprocessing as mp
    over = dict(d_model=64, n_heads=4, n_layers=2, **NODROP)
    res = {}
    for graphs in (True, False):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        mp.spawn(_dp_worker, args=(2, port, str(tmp_path), over, graphs), nprocs=2, join=True)
        res[graphs] = [torch.load(tmp_path / f"g{int(graphs)}_rank{r}.pt") for r in (0, 1)]
    g0, g1 = res[True]
    e0, _ = res[False]
    info = g0["info"]
    # forward: one SyncBN exchange per layer -> n_layers + 1 segments; backward: per layer one SyncBN exchange + the layer's hook
    assert info and info[0]["fwd_host_calls"] == 2 and info[0]["bwd_host_calls"] >= 2 * 2 + 2, info
    for a, b in zip(g0["flat"], g1["flat"]):
        assert torch.equal(a, b)
    assert torch.equal(g0["bn"], g1["bn"])
    for a, b in zip(g0["losses"], e0["losses"]):
        assert abs(a - b) <= 1e-4 * abs(b), (g0["losses"], e0["losses"])
    for a, b in zip(g0["flat"], e0["flat"]):
        assert (a - b).norm() <= 3e-3 * b.norm()


def test_bf16_gradient_wire_follows_the_fp32_exchange(tmp_path):
    """MI355X_GRAD_WIRE=bf16 (GradSync(wire_dtype=bfloat16), the analogue of DDP's bf16_compress_hook): buckets are scaled by
    1/world, rounded to bf16, all-reduced and widened back; the optimizer then applies no further scale.  The replicas stay
    identical and the loss curve follows the fp32 exchange within bf16 rounding of the gradients."""
    import torch.multiprocessing as mp
    over = dict(d_model=64, n_heads=4, n_layers=2, **NODROP)
    res = {}
    for wire in ("bf16", "fp32"):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        mp.spawn(_dp_worker, args=(2, port, str(tmp_path), over, True, wire), nprocs=2, join=True)
        tag = "" if wire == "fp32" else wire
        res[wire] = [torch.load(tmp_path / f"g1{tag}_rank{r}.pt") for r in (0, 1)]
    b0, b1 = res["bf16"]
    f0, _ = res["fp32"]
    for a, b in zip(b0["flat"], b1["flat"]):
        assert torch.equal(a, b)
    assert b0["losses"][-1] < 0.98 * b0["losses"][0]
    for a, b in zip(b0["losses"], f0["losses"]):
        assert abs(a - b) <= 5e-3 * abs(b), (b0["losses"], f0["losses"])


def test_auto_mode_times_both_ways_and_keeps_one():
    """the default (MI355X_GRAPHS=auto): a shape runs eagerly for its first steps, is recorded, replayed for a few timed steps and
    then stays on whichever was faster on the device; the optimisation trajectory is the eager one either way"""
    over = dict(d_model=64, n_heads=4, n_layers=2, **NODROP)
    batches = [_batch()]
    torch.manual_seed(21)
    m = _model(dict(over, compute_dtype=torch.float32)).to(dev).train()
    m.decoder.compute_dtype = torch.float32
    assert m.encoder.use_graphs and m.encoder.graph_auto  # the defaults
    m.setup_optimization(dict(name="adamw", lr=1e-3, betas=[0.9, 0.98], weight_decay=1e-3))
    losses = []
    for i in range(16):
        losses.append(m.fit_step(batches[0])["loss"].item())  # (.item() drains the queue: the end events are reached)
    assert m.encoder.graphs_settled()
    info = m.encoder.graph_info()
    assert len(info) == 1 and str(info[0]["decided"]).split()[0] in ("eager", "graph") and info[0]["auto"] is not None, info
    m_e, l_e = _run(over, False, 16, batches)
    _same_trajectory(losses, l_e)


def test_many_padded_lengths_share_the_arena_and_stay_the_eager_sequence():
    """a duration-shaped loader with the featurizer's `pad_to` (features.py:501): five padded lengths, visited in turn, each
    recorded once (forced replay) -- every length's launches point into the ONE step arena, so a replay of length A after a live or
    replayed step of length B must still be the eager step of A.  Lengths after the first recorded one need a single live visit."""
    over = dict(d_model=64, n_heads=4, n_layers=2, **NODROP)
    secs = [1.0, 1.3, 0.8, 1.6, 1.15]
    batches = [_batch(B=3, secs=s, seed=8 + i, lens=[int(16000 * s), int(12000 * s), int(9000 * s)]) for i, s in enumerate(secs)]
    order = [0, 1, 2, 3, 4] * 5 + [4, 2, 0, 3, 1, 1, 3]

    def run(graphs):
        torch.manual_seed(21)
        m = _model(dict(over, compute_dtype=torch.float32)).to(dev).train()
        m.decoder.compute_dtype = torch.float32
        m.preprocessor.featurizer.pad_to = 16
        m.encoder.use_graphs, m.encoder.graph_auto = graphs, False
        m.setup_optimization(dict(name="adamw", lr=1e-3, betas=[0.9, 0.98], weight_decay=1e-3))
        ls = [m.fit_step(batches[i])["loss"].item() for i in order]
        torch.cuda.synchronize()
        return m, ls

    m_g, l_g = run(True)
    m_e, l_e = run(False)
    info = m_g.encoder.graph_info()
    assert len(info) == len({-(-(b[0].shape[1] // 160 + 1) // 16) * 16 for b in batches}) == 5, info
    _check_replay_kind(info, "tape")
    assert m_g.encoder.graphs_recorded()
    # 5 lengths x 2 warm-up visits + the recording visits are the only live ones; everything after replays
    assert m_g.encoder.replayed_steps >= len(order) - 3 * 5 and m_g.encoder.live_steps <= 2 * 5, (m_g.encoder.replayed_steps, m_g.encoder.live_steps)
    _same_trajectory(l_g, l_e)


def test_auto_mode_decides_once_per_configuration_not_once_per_padded_length():
    """MI355X_GRAPHS=auto with several padded lengths: the first length that finishes its live-vs-recorded trial decides; the other
    lengths take the decision over (no trial of their own) and the trajectory stays the eager one"""
    over = dict(d_model=64, n_heads=4, n_layers=2, **NODROP)
    batches = [_batch(B=3, secs=s, seed=8 + i) for i, s in enumerate([1.0, 1.4, 0.7])]
    order = [0] * 13 + [1, 2, 1, 2, 1, 2, 0, 1, 2]
    torch.manual_seed(21)
    m = _model(dict(over, compute_dtype=torch.float32)).to(dev).train()
    m.decoder.compute_dtype = torch.float32
    m.setup_optimization(dict(name="adamw", lr=1e-3, betas=[0.9, 0.98], weight_decay=1e-3))
    losses = [m.fit_step(batches[i])["loss"].item() for i in order]
    enc = m.encoder
    sets = list(enc._graph_sets.values())
    assert len(sets) == 3 and all(g.decided is not None for g in sets), [g.decided for g in sets]
    assert [g.inherited for g in sets].count(False) == 1 and len({g.decided for g in sets}) == 1
    assert enc.graphs_settled() and enc.graphs_recorded()
    m_e, l_e = _run(over, False, 0, batches)
    l_e = [m_e.fit_step(batches[i])["loss"].item() for i in order]
    _same_trajectory(losses, l_e)
