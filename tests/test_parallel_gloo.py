"""N > 1 path on CPU: world_size-2 `gloo` run of the bucketed gradient exchange (nemo_amd/parallel.py) -- ranges arrive
in reverse-layer order like the backward sequencer produces them, are merged into buckets, all-reduced asynchronously,
and `wait()` returns the 1/world factor the fused AdamW applies."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nemo_amd.parallel import GradSync
        n = 64 * 50
        grad = torch.arange(n, dtype=torch.float32) * (rank + 1)
        gs = GradSync(grad, bucket_bytes=64 * 4 * 12)  # 12 "parameters" of 64 floats per bucket
        launched_before_wait = 0
        # reverse-layer order: layer i owns [64*5*i, 64*5*(i+1))
        for layer in range(9, -1, -1):
            gs.ready(64 * 5 * layer, 64 * 5 * (layer + 1))
            launched_before_wait = max(launched_before_wait, len(gs.reduced_ranges()))
        scale = gs.wait()
        expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
        ok = torch.equal(grad, expect) and abs(scale - 1.0 / world) < 1e-12 and launched_before_wait >= 2
        # a second step reuses the object
        grad.fill_(float(rank))
        gs.ready(0, n)
        gs.wait()
        ok = ok and torch.equal(grad, torch.full((n,), float(sum(range(world)))))
        # ranges covering more than the buffer between two wait() calls (a second backward pass) are refused, not silently
        # reduced twice
        gs.ready(0, n // 2)
        try:
            gs.ready(0, n)
            ok = False
        except RuntimeError as e:
            ok = ok and "between two wait()" in str(e)
        gs.wait()
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_bucketed_grad_sync_world2_gloo():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _worker_bf16_wire(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nemo_amd.parallel import GradSync
        n = 64 * 40
        g = torch.Generator().manual_seed(100 + rank)
        grad = torch.randn(n, generator=g)
        mine = grad.clone()
        gs = GradSync(grad, bucket_bytes=64 * 4 * 8, wire_dtype=torch.bfloat16)
        for layer in range(7, -1, -1):
            gs.ready(64 * 5 * layer, 64 * 5 * (layer + 1))
        scale = gs.wait()
        # what every rank must hold: sum over ranks of bf16(g_r / world), accumulated in bf16 by the collective -- i.e. the mean
        # to bf16 precision, already scaled (grad_scale 1), identical bits on all ranks
        others = [torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
        mean = sum(others) / world
        err = (grad * scale - mean).abs().max().item()
        gathered = [torch.empty_like(grad) for _ in range(world)]
        dist.all_gather(gathered, grad)
        ret[rank] = bool(scale == 1.0 and err < 2e-2 and all(torch.equal(gathered[0], t) for t in gathered)
                         and torch.equal(mine, others[rank]))
    finally:
        dist.destroy_process_group()


def test_bf16_gradient_wire_world2_gloo():
    """GradSync(wire_dtype=bfloat16): buckets are pre-scaled by 1 / world, rounded to bf16, all-reduced and widened back; the
    optimizer's remaining factor is 1 and every rank ends with the same bits"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_bf16_wire, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _torch_adamw(p, g, m, v, lr, b1, b2, eps, wd, step, grad_scale=1.0, clip_coef=None, ema=None, ema_decay=0.0):
    """host stand-in for the HIP launch (same update rule, adamw_kernel in csrc/optim.hip) so the slice bookkeeping of
    the optimizer-behind-backward path can run on CPU"""
    gr = g * grad_scale
    p.mul_(1 - lr * wd)
    m.mul_(b1).add_(gr, alpha=1 - b1)
    v.mul_(b2).addcmul_(gr, gr, value=1 - b2)
    p.sub_((lr / (1 - b1 ** step)) * m / (v.sqrt() / (1 - b2 ** step) ** 0.5 + eps))
    if ema is not None:
        ema.mul_(ema_decay).add_(p, alpha=1 - ema_decay)


def _worker_opt(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nemo_amd import ops
        from nemo_amd.flat import FlatParams
        from nemo_amd.optim import FusedAdamW
        from nemo_amd.parallel import GradSync
        ops.adamw_step = _torch_adamw
        torch.manual_seed(0)
        layers = 6
        mod = torch.nn.Sequential(*[torch.nn.Linear(32, 32) for _ in range(layers)])
        ref = torch.nn.Sequential(*[torch.nn.Linear(32, 32) for _ in range(layers)])
        ref.load_state_dict(mod.state_dict())
        fp = FlatParams(mod); fp.build(torch.device("cpu"))
        opt = FusedAdamW([fp], lr=1e-2, betas=(0.9, 0.98), weight_decay=1e-2)
        ropt = torch.optim.AdamW(ref.parameters(), lr=1e-2, betas=(0.9, 0.98), weight_decay=1e-2)
        gs = GradSync(fp.grad, bucket_bytes=4 * 2200)  # ~2 layers per bucket
        calls = []
        gs.after_reduce = lambda lo, hi: (calls.append((lo, hi)), opt.step_range(fp, lo, hi))
        for step in range(3):
            fp.grad.zero_(); ropt.zero_grad()  # (FusedAdamW.zero_grad is a HIP fill)
            assert opt.begin_step(lr=1e-2, grad_scale=1.0 / world)
            gens = [torch.Generator().manual_seed(100 * step + r) for r in range(world)]
            per_rank = [[torch.randn(p.shape, generator=g) for p in ref.parameters()] for g in gens]
            for q, *gr in zip(ref.parameters(), *per_rank):
                q.grad = sum(gr) / world
            mine = dict(zip([n for n, _ in mod.named_parameters()], per_rank[rank]))
            for i in range(layers - 1, -1, -1):  # reverse-layer order, like the backward sequencer
                for n, p in mod.named_parameters():
                    if n.startswith(f"{i}."):
                        p.grad.copy_(mine[n])
                gs.ready(*fp.range_of(f"{i}."))
            n_before_wait = len(calls)
            gs.wait()
            opt.finish_step()
            ropt.step()
            assert n_before_wait >= 2  # slices were updated while "backward" was still producing gradients
        err = max((p.detach() - q.detach()).abs().max().item() for p, q in zip(mod.parameters(), ref.parameters()))
        flat = fp.flat.clone()
        dist.all_reduce(flat)
        same = torch.allclose(flat, fp.flat * world, rtol=0, atol=1e-6)  # the replicas stay in lock-step
        ret[rank] = bool(err < 1e-5 and same and opt.step_count == 3)
    finally:
        dist.destroy_process_group()


def test_optimizer_behind_the_exchange_world2_gloo():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_opt, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _worker_data(rank, world, port, ret, manifest, vocab):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nemo_amd.models import EncDecCTCModel, conformer_ctc_config
        cfg = conformer_ctc_config("small", vocab_size=len(vocab), d_model=32, n_heads=2, n_layers=1)
        cfg["labels"] = vocab
        model = EncDecCTCModel(cfg)
        ok = model.world_size == world
        out = {}
        for name, extra in (("ssb", dict(use_semi_sorted_batching=True, semi_sort_synced_rng=True)), ("ddp", dict())):
            dl = model.setup_training_data(dict(manifest_filepath=manifest, batch_size=4, return_sample_id=True,
                                                shuffle=True, **extra))
            ids, lens = [], []
            for sig, sl, tok, tl, sid in dl:
                ids.append(sid.tolist()); lens.append(int(sl.max()))
            gathered = [None] * world
            dist.all_gather_object(gathered, (ids, lens))
            out[name] = gathered
        n = 26
        ssb_ids = [i for r in out["ssb"] for b in r[0] for i in b]
        ok = ok and set(ssb_ids) == set(range(n)) and len(ssb_ids) - n < world * 4
        ok = ok and len(out["ssb"][0][0]) == len(out["ssb"][1][0])          # same number of steps
        # step k of both ranks is padded to (nearly) the same length
        ok = ok and all(abs(a - b) <= 0.15 * max(a, b) for a, b in zip(out["ssb"][0][1], out["ssb"][1][1]))
        ddp_ids = [i for r in out["ddp"] for b in r[0] for i in b]
        ok = ok and set(ddp_ids) == set(range(n)) and len(ddp_ids) == n  # DistributedSampler: 13 each
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_training_data_is_partitioned_across_ranks_world2_gloo(tmp_path):
    import json
    import wave
    import numpy as np
    vocab = [" "] + list("abcdefg")
    rs = np.random.RandomState(1)
    with open(tmp_path / "m.json", "w") as mf:
        for i in range(26):
            ns = int(rs.randint(1600, 16000))
            with wave.open(str(tmp_path / f"u{i}.wav"), "wb") as f:
                f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
                f.writeframes((rs.randn(ns) * 1000).astype(np.int16).tobytes())
            mf.write(json.dumps(dict(audio_filepath=f"u{i}.wav", duration=ns / 16000, text="ab cd")) + "\n")
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_data, args=(world, _free_port(), ret, str(tmp_path / "m.json"), vocab), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _worker_hostsum(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nemo_amd.parallel import HostSum
        hs = HostSum()
        ret[rank] = [hs(100.0 + rank), hs(7.0 * (rank + 1))]
    finally:
        dist.destroy_process_group()


def test_host_sum_of_ragged_syncbn_counts_world2_gloo():
    """the SyncBatchNorm element count with ragged ranks: per-rank B*T' summed on the host (no GPU round trip)"""
    from nemo_amd.parallel import HostSum
    assert HostSum()(5.0) == 5.0  # no process group: identity
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_hostsum, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: [201.0, 21.0], 1: [201.0, 21.0]}


def test_grad_sync_is_a_noop_for_world1():
    from nemo_amd.parallel import GradSync
    g = torch.ones(128)
    gs = GradSync(g)
    gs.ready(0, 128)
    assert gs.wait() == 1.0 and torch.equal(g, torch.ones(128))


def _worker_tail(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nemo_amd.parallel import GradSync
        per = 64 * 5
        n = per * 21
        grad = torch.arange(n, dtype=torch.float32) * (rank + 1)
        # buckets of 8 layers, tail of 2 layers: 21 layers arrive top-down -> [20..13] and [12..5] as full buckets, then -- when only
        # layers 1 and 0 are still to come -- the early cut [4..2], and the tail [1..0] the moment the buffer is complete
        gs = GradSync(grad, bucket_bytes=per * 4 * 8, tail_bytes=per * 4 * 2)
        seen = []
        for layer in range(20, -1, -1):
            gs.ready(per * layer, per * (layer + 1))
            seen.append(list(gs.reduced_ranges()))
        want = [(per * 13, per * 21), (per * 5, per * 13), (per * 2, per * 5), (0, per * 2)]
        ok = seen[-1] == want                       # complete BEFORE wait(): nothing is left for the end of backward
        ok = ok and seen[-2] == want[:3] and seen[-3] == want[:3] and seen[-4] == want[:2]
        gs.wait()
        ok = ok and gs.launches_last_step == 4
        ok = ok and torch.equal(grad, torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world)))
        # next step: the bookkeeping starts over (same cuts again); a buffer that fits one bucket goes out when its LAST range arrives
        grad.copy_(torch.arange(n, dtype=torch.float32) * (rank + 1))
        for layer in range(20, -1, -1):
            gs.ready(per * layer, per * (layer + 1))
        ok = ok and list(gs.reduced_ranges()) == want
        gs.wait()
        small = torch.full((per,), float(rank + 1))
        g2 = GradSync(small, bucket_bytes=64 << 20)
        g2.ready(64, per)
        ok = ok and g2.reduced_ranges() == []
        g2.ready(0, 64)
        ok = ok and g2.reduced_ranges() == [(0, per)]
        g2.wait()
        ok = ok and torch.equal(small, torch.full((per,), float(sum(r + 1 for r in range(world)))))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_tail_bucket_is_cut_early_and_complete_buffers_leave_at_once_world2_gloo():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_tail, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _worker_mailbox_agreement(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nemo_amd.mailbox import StatsMailbox
        # no GPU on this box: no rank can allocate a mailbox -- every rank still walks through BOTH rounds of the agreement
        # (all_gather of the handles, MIN over the open results) and gets None, and the process group is still usable afterwards
        mb = StatsMailbox.create(torch.device("cuda:0") if rank == 0 else torch.device("cpu"), n_max=1025)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t)
        ret[rank] = (mb is None) and float(t[0]) == 3.0
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.is_available(), reason="the no-GPU side of StatsMailbox.create (the GPU side: tests/test_mailbox_gpu.py)")
def test_mailbox_creation_is_all_or_nothing_across_ranks_world2_gloo():
    """nemo_amd/mailbox.py: a rank that cannot allocate / export / map a mailbox must not leave the others with one -- the ranks
    agree through the process group and ALL fall back to it (SyncBatchNorm exchanges stay on torch.distributed.all_reduce)"""
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker_mailbox_agreement, args=(2, port, ret), nprocs=2, join=True)
    assert ret[0] and ret[1]


def test_bucketed_exchange_tail_cut_and_bf16_wire_world4_gloo():
    """the same bookkeeping with FOUR ranks (the scaling runs use 2, 4 and 8): bucket merge order, the early tail cut, sums over
    four contributions, the 1 / world factor, and the bf16 wire's pre-scale by 1 / 4"""
    world = 4
    mgr = mp.Manager()
    for worker in (_worker, _worker_tail, _worker_bf16_wire):
        ret = mgr.dict()
        mp.spawn(worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert dict(ret) == {r: True for r in range(world)}, worker.__name__
