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
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_bucketed_grad_sync_world2_gloo():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_grad_sync_is_a_noop_for_world1():
    from nemo_amd.parallel import GradSync
    g = torch.ones(128)
    gs = GradSync(g)
    gs.ready(0, 128)
    assert gs.wait() == 1.0 and torch.equal(g, torch.ones(128))
