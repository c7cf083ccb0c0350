"""Statistics mailbox (csrc/mailbox.hip, nemo_amd/mailbox.py): the SyncBatchNorm exchange as one kernel launch over peer-mapped
memory.  Two processes share the one GPU of the test box (hipIpc handles work between processes on the same device), gloo
carries the handles and the reference all-reduce: for two ranks a + b is the same number in either order, so the mailbox sum has
to equal the process group's all-reduce BIT FOR BIT."""
import json
import os
import socket
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {"available": False}
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        from nemo_amd.mailbox import StatsMailbox
        mb = StatsMailbox.create(dev, n_max=8193, timeout_ms=10000)
        if mb is not None:
            res["available"] = True
            res["kind"] = mb.kind
            bad = []
            side = torch.cuda.Stream()
            for k in range(48):
                n = (1025, 1, 8193, 513, 2049)[k % 5]
                g = torch.Generator().manual_seed(1000 * k + rank)
                x = torch.randn(n, dtype=torch.float64, generator=g) * (10.0 ** (k % 7 - 3))
                ref = x.clone()
                dist.all_reduce(ref)                       # gloo, host
                y = x.to(dev)
                if k % 6 == 5 and rank == 1:
                    time.sleep(0.05)                       # a late rank: the early one's kernel waits on the flag
                if k % 2:                                  # the exchange runs on whatever stream is current
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        mb.all_reduce_(y)
                    torch.cuda.current_stream().wait_stream(side)
                else:
                    mb.all_reduce_(y)
                torch.cuda.synchronize()
                if not torch.equal(y.cpu(), ref):
                    bad.append((k, float((y.cpu() - ref).abs().max())))
            # back-to-back exchanges without a host sync in between (slot reuse: 4 slots, 12 exchanges in flight order)
            ys, refs = [], []
            for k in range(12):
                g = torch.Generator().manual_seed(77 * k + rank)
                x = torch.randn(1025, dtype=torch.float64, generator=g)
                r = x.clone(); dist.all_reduce(r); refs.append(r)
                ys.append(x.to(dev))
            torch.cuda.synchronize()
            for y in ys:
                mb.all_reduce_(y)
            torch.cuda.synchronize()
            for k, (y, r) in enumerate(zip(ys, refs)):
                if not torch.equal(y.cpu(), r):
                    bad.append((100 + k, float((y.cpu() - r).abs().max())))
            done, missing, kind = mb.status()
            res.update(bad=bad, done=done, missing=missing)
            with pytest.raises(ValueError):
                mb.all_reduce_(torch.zeros(8194, dtype=torch.float64, device=dev))
            with pytest.raises(ValueError):
                mb.all_reduce_(torch.zeros(8, dtype=torch.float32, device=dev))
            mb.close()
    finally:
        with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
            json.dump(res, f)
        dist.destroy_process_group()


def test_mailbox_sum_equals_the_process_group_all_reduce_bit_for_bit(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [json.load(open(tmp_path / f"rank{i}.json")) for i in range(2)]
    if not all(x["available"] for x in r):
        assert not any(x["available"] for x in r)  # all-or-nothing across the ranks
        pytest.skip("hipIpc handles cannot be exported / mapped on this box: the encoders stay on the process group")
    for x in r:
        assert x["bad"] == [] and x["missing"] == 0 and x["done"] == 48 + 12, x


def _worker_timeout(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {"available": False}
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        from nemo_amd.mailbox import StatsMailbox
        mb = StatsMailbox.create(dev, n_max=64, timeout_ms=400)
        if mb is not None:
            res["available"] = True
            y = torch.full((33,), float(rank + 1), dtype=torch.float64, device=dev)
            mb.all_reduce_(y)
            torch.cuda.synchronize()
            res["first"] = y.cpu().tolist()
            assert mb.poll() == 0 or True   # first poll only enqueues its copy
            dist.barrier()
            if rank == 0:   # the peer never comes to this exchange
                z = torch.full((33,), 5.0, dtype=torch.float64, device=dev)
                t0 = time.time()
                mb.all_reduce_(z)
                torch.cuda.synchronize()
                res["timeout_s"] = time.time() - t0
                res["nan_after_timeout"] = bool(torch.isnan(z).all())
                done, missing, _ = mb.status()
                res["done"], res["missing"] = done, missing
                z2 = torch.full((33,), 7.0, dtype=torch.float64, device=dev)
                t0 = time.time()
                mb.all_reduce_(z2)          # latched: returns at once, poisoned -- never the local sums
                torch.cuda.synchronize()
                res["latched_s"] = time.time() - t0
                res["nan_when_latched"] = bool(torch.isnan(z2).all())
                raised = None
                for _ in range(3):          # non-blocking: the copy enqueued by one poll is read by the next
                    try:
                        mb.poll()
                    except RuntimeError as e:
                        raised = str(e)
                        break
                    torch.cuda.synchronize()
                res["poll_raised"] = raised
            else:
                time.sleep(1.5)
            mb.close()
    finally:
        with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
            json.dump(res, f)
        dist.destroy_process_group()


def test_a_missing_peer_poisons_the_result_and_the_step_loop_hears_about_it(tmp_path):
    """on a peer time-out the kernel must not hand back the local sums nor advance its sequence number, and the
    training loop has to learn about the latch without a blocking copy (StatsMailbox.poll, called once per fit_step)."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker_timeout, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [json.load(open(tmp_path / f"rank{i}.json")) for i in range(2)]
    if not all(x["available"] for x in r):
        pytest.skip("hipIpc handles cannot be exported / mapped on this box")
    assert r[0]["first"] == [3.0] * 33 and r[1]["first"] == [3.0] * 33
    x = r[0]
    assert x["nan_after_timeout"] and x["nan_when_latched"], x
    assert x["missing"] == 2 and x["done"] == 1, x          # 1 + the rank that never arrived; the sequence did not advance
    assert 0.3 <= x["timeout_s"] <= 5.0 and x["latched_s"] < 0.3, x
    assert x["poll_raised"] and "rank 1 never arrived" in x["poll_raised"], x


def test_mailbox_needs_a_process_group_and_refuses_bad_arguments():
    import ctypes as C
    from nemo_amd._lib import lib
    from nemo_amd.mailbox import StatsMailbox
    assert StatsMailbox.create(torch.device("cuda:0")) is None  # no process group: nothing to exchange with
    mb, h = C.c_void_p(), (C.c_ubyte * 64)()
    assert lib.mi355x_mailbox_create(0, 0, 16, 0, 0, C.byref(mb), h) == 1
    assert lib.mi355x_mailbox_create(2, 2, 16, 0, 0, C.byref(mb), h) == 1
    assert lib.mi355x_mailbox_create(2, 0, 16, 0, 7, C.byref(mb), h) == 1
    # a one-rank mailbox is its own peer: the exchange is the identity (and works without any IPC mapping)
    rc = lib.mi355x_mailbox_create(1, 0, 64, 0, 0, C.byref(mb), h)
    if rc != 0:
        pytest.skip(f"no exportable device memory on this box (rc = {rc})")
    x = torch.arange(64, dtype=torch.float64, device="cuda:0")
    for _ in range(9):
        assert lib.mi355x_mailbox_exchange(mb, x.data_ptr(), 64, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(x.cpu(), torch.arange(64, dtype=torch.float64))
    st = (C.c_longlong * 3)()
    assert lib.mi355x_mailbox_status(mb, st) == 0 and st[0] == 9 and st[1] == 0
    assert lib.mi355x_mailbox_exchange(mb, x.data_ptr(), 65, None) == 1
    lib.mi355x_mailbox_destroy(mb)
    # a two-rank mailbox whose peer was never opened refuses to launch
    rc = lib.mi355x_mailbox_create(2, 0, 64, 0, 0, C.byref(mb), h)
    assert rc == 0
    assert lib.mi355x_mailbox_exchange(mb, x.data_ptr(), 64, None) == 1
    lib.mi355x_mailbox_destroy(mb)
