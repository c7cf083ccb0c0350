"""Statistics mailbox: the SyncBatchNorm exchanges of the data-parallel step as one kernel launch each (csrc/mailbox.hip).

The reference gets these exchanges from `torch.nn.SyncBatchNorm` (`trainer.sync_batchnorm: true`,
examples/asr/conf/conformer/conformer_ctc_bpe.yaml:209, over the BatchNorm1d of parts/submodules/conformer_modules.py:339):
one `all_reduce` of the raw sums per layer forward and one per layer backward, 36 latency-bound 8-KB collectives per
Conformer-CTC-Large step, on the same RCCL stream as the 64-MiB gradient buckets.  A `StatsMailbox` keeps them off the process
group: every rank allocates a mailbox in its own HBM, the ranks exchange the hipIpc handles ONCE (one `all_gather` over the
job's process group -- the only collective here) and map each other's mailboxes; after that an exchange is
`mi355x_mailbox_exchange` on the compute stream (peer stores over xGMI + sequence flags + a rank-ordered sum, so every rank
holds bit-identical results).  `StatsMailbox.create()` is all-or-nothing across the ranks: if any rank cannot export or map
a mailbox (no IPC in this sandbox, a peer on another node), every rank gets `None` and the caller stays on the process group.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
import torch.distributed as dist

from . import _lib
from ._lib import lib

HANDLE_BYTES = 64


class StatsMailbox:
    def __init__(self, mb, world: int, rank: int, n_max: int, kind: int, device):
        self._mb, self.world, self.rank, self.n_max, self.kind, self.device = mb, world, rank, n_max, kind, device
        self.exchanges = 0

    # ------------------------------------------------------------------ construction (collective)
    @classmethod
    def create(cls, device, group=None, n_max: int = 8193, timeout_ms: int = 10000) -> Optional["StatsMailbox"]:
        """Collective over `group`: every rank calls it at the same point.  Returns a mailbox on every rank or None on every rank.
        `timeout_ms`: how long an exchange waits for a peer's flag before it latches the error word (10 s: far beyond any skew between
        healthy ranks -- two processes time-sliced on ONE GPU in the rehearsals hand over in ~0.3 s, profiles/r4_mailbox.md -- and far
        below a collective watchdog)."""
        if not (dist.is_available() and dist.is_initialized()):
            return None
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        if world < 2:
            return None
        device = torch.device(device)
        on_host = dist.get_backend(group) == "gloo"  # (gloo carries host tensors; RCCL carries device tensors)
        have_gpu = device.type == "cuda" and torch.cuda.is_available()
        # Do all ranks sit on ONE physical device (the two-ranks-on-one-GPU rehearsals)?  Only then may the second attempt fall
        # back to plain coarse-grained hipMalloc memory: across devices a peer's xGMI stores into coarse-grained memory are not
        # guaranteed visible to a kernel already running against its XCD's L2, whatever the scope of the atomics.
        # Both the UUID and the (host, device index, PCI bus id) must agree: a runtime may hand out one UUID string for distinct
        # devices (bench.py stopped trusting UUIDs alone for that reason), and the fallback is unsafe across devices -- fail closed.
        uuid = [0] * 16
        where = [0] * 8
        if have_gpu:
            try:
                props = torch.cuda.get_device_properties(device)
                uuid = list(props.uuid.bytes)
                import socket
                import zlib
                idx = device.index if device.index is not None else torch.cuda.current_device()
                bus = (int(getattr(props, "pci_domain_id", 0)), int(getattr(props, "pci_bus_id", -1 - rank)),
                       int(getattr(props, "pci_device_id", 0)))
                where = [zlib.crc32(socket.gethostname().encode()) & 0x7FFFFFFF, idx, *bus, 0, 0, 0]
            except Exception:  # noqa: BLE001 -- no identity on this build: treat every rank as its own device
                uuid = [rank + 1] * 16
                where = [rank + 1] * 8
        mine_u = torch.tensor(uuid + where, dtype=torch.int32)
        if not on_host:
            mine_u = mine_u.to(device)
        all_u = [torch.empty_like(mine_u) for _ in range(world)]
        dist.all_gather(all_u, mine_u, group=group)
        one_device = all(torch.equal(u.cpu(), all_u[0].cpu()) for u in all_u)
        for mem_kind in ((0, 3) if one_device else (0,)):  # first: the best exportable kind per rank; second (one device only): plain memory
            mb = C.c_void_p()
            handle = (C.c_ubyte * HANDLE_BYTES)()
            rc = -1  # (a rank without a GPU still takes part in the agreement below, so that every rank gets the same answer)
            if have_gpu:
                with torch.cuda.device(device):
                    rc = lib.mi355x_mailbox_create(world, rank, n_max, timeout_ms, mem_kind, C.byref(mb), handle)
            mine = torch.tensor([1 if rc == 0 else 0] + list(handle), dtype=torch.int32)
            if not on_host:
                mine = mine.to(device)
            every = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(every, mine, group=group)
            every = [t.cpu() for t in every]
            ok = rc == 0 and all(int(t[0]) == 1 for t in every)
            if ok:
                with torch.cuda.device(device):
                    for r in range(world):
                        if r == rank:
                            continue
                        h = (C.c_ubyte * HANDLE_BYTES)(*[int(v) for v in every[r][1:]])
                        if lib.mi355x_mailbox_open(mb, r, h) != 0:
                            ok = False
                            break
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
            if not on_host:
                flag = flag.to(device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if int(flag.item()) == 1:
                st = (C.c_longlong * 3)()
                lib.mi355x_mailbox_status(mb, st)
                return cls(mb, world, rank, n_max, int(st[2]), device)
            if rc == 0:
                lib.mi355x_mailbox_destroy(mb)
        return None

    # ------------------------------------------------------------------ the exchange
    def all_reduce_(self, stats: torch.Tensor) -> torch.Tensor:
        """stats (f64, contiguous, on this mailbox's device) <- sum over the ranks, in place, on the current stream"""
        if stats.dtype != torch.float64 or not stats.is_cuda or not stats.is_contiguous():
            raise ValueError("StatsMailbox.all_reduce_: a contiguous float64 device tensor")
        if stats.numel() > self.n_max:
            raise ValueError(f"StatsMailbox.all_reduce_: {stats.numel()} values, the mailbox was sized for {self.n_max}")
        _lib.check(lib.mi355x_mailbox_exchange(self._mb, stats.data_ptr(), stats.numel(),
                                               torch.cuda.current_stream(stats.device).cuda_stream), "mi355x_mailbox_exchange")
        self.exchanges += 1
        return stats

    def poll(self, stream=None):
        """non-blocking latch check for the training loop (once per step): raises as soon as a previous exchange is known to
        have timed out on this rank (the result of that exchange is NaN by construction, csrc/mailbox.hip)"""
        st = (C.c_longlong * 4)()
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        _lib.check(lib.mi355x_mailbox_poll(self._mb, s.cuda_stream, st), "mi355x_mailbox_poll")
        if int(st[1]):
            raise RuntimeError(f"StatsMailbox: rank {int(st[1]) - 1} never arrived at a SyncBatchNorm exchange within the time-out "
                               f"(after {int(st[0])} completed exchanges); this rank's statistics of that step are NaN. "
                               "Restart the job (or run without MI355X_SYNCBN_MAILBOX to stay on the process group).")
        return int(st[0])

    def status(self):
        """(exchanges completed on the device, 0 or 1 + the rank that never arrived, memory kind) -- blocks"""
        st = (C.c_longlong * 3)()
        _lib.check(lib.mi355x_mailbox_status(self._mb, st), "mi355x_mailbox_status")
        return int(st[0]), int(st[1]), int(st[2])

    def close(self, group=None):
        """collective: no rank unmaps while a peer may still store into its mailbox"""
        if self._mb is not None:
            torch.cuda.synchronize(self.device)
            if dist.is_available() and dist.is_initialized():
                dist.barrier(group=group)
            lib.mi355x_mailbox_destroy(self._mb)
            self._mb = None
