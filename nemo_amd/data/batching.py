"""Duration-aware batch shaping for data-parallel training: `SemiSortBatchSampler`
(`nemo/collections/asr/parts/utils/asr_batching.py:27-204`, Ge et al. 2021).

Utterances are ordered by `duration + U(-b, b)` with `b = (max - min) * randomization_factor / 2`, the ordered list is cut
into micro-batches, padded with randomly repeated utterances to a multiple of the world size, and dealt out by
`order[rank::world]` -- so the k-th batch of every rank holds neighbours of the same stretch of the ordering: all ranks
step over (nearly) equal padded lengths, which is what keeps the weak-scaling efficiency of a padded batch (the RCCL
all-reduce waits for the slowest rank).  Batches are then visited in an order drawn from `seed + epoch + 1`.

Randomness: the reference draws the noise / the dropped tail / the padding from numpy's GLOBAL generator in exactly this
order (`uniform`, [`choice`], [`randint`]), which only gives every rank the same ordering when the global generator was
seeded identically on all ranks (Lightning's `seed_everything`).  The same three calls are made here on a generator
object: by default `np.random` itself (drop-in behaviour, pinned against the reference class under equal seeds), or a
private `np.random.RandomState(seed + epoch)` with `synced_rng=True`, which makes the cross-rank agreement a property of the
sampler instead of the launcher.
"""
from __future__ import annotations

import math
from typing import Iterator, List, Optional, Sequence

import numpy as np
import torch


class SemiSortBatchSampler:
    def __init__(self, global_rank: int, world_size: int, durations: Sequence[float], batch_size: int,
                 batch_shuffle: bool = True, drop_last: bool = False, randomization_factor: Optional[float] = None,
                 seed: int = 42, synced_rng: bool = False) -> None:
        if randomization_factor is None:
            randomization_factor = 0.1
        if randomization_factor < 0.0:
            raise ValueError(f"Randomization factor must be non-negative but found {randomization_factor}.")
        self.rank, self.num_replicas = global_rank, world_size
        self.durations = np.array(durations, dtype=np.float32)
        self.shuffle, self.micro_batch_size, self.drop_last = batch_shuffle, batch_size, drop_last
        self.epoch, self.seed = 0, seed
        self.randomization_factor = randomization_factor
        self.synced_rng = synced_rng
        self.local_num_batches = self._calculate_local_num_batches()

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def _calculate_local_num_batches(self) -> int:
        n = len(self.durations)
        if self.drop_last:
            n -= n % self.micro_batch_size
        global_num_batches = math.ceil(n / self.micro_batch_size)
        global_num_batches += (self.num_replicas - global_num_batches % self.num_replicas) % self.num_replicas
        return global_num_batches // self.num_replicas

    def _make_batches(self) -> List[np.ndarray]:
        rng = np.random.RandomState(self.seed + self.epoch) if self.synced_rng else np.random
        if len(self.durations) == 0:
            return []
        bound = (float(np.max(self.durations)) - float(np.min(self.durations))) * self.randomization_factor / 2
        noise = rng.uniform(low=-bound, high=bound, size=len(self.durations))
        order = np.argsort(self.durations + noise)
        if self.drop_last:
            tail = len(order) % self.micro_batch_size
            order = np.delete(order, rng.choice(len(order), tail, replace=False))
        global_num_batches = math.ceil(len(order) / self.micro_batch_size)
        if global_num_batches == 0:
            return []
        pad_batches = (self.num_replicas - global_num_batches % self.num_replicas) % self.num_replicas
        if pad_batches:
            extra = rng.randint(low=0, high=len(order), size=pad_batches * self.micro_batch_size)
            order = np.concatenate((order, order[extra]), axis=0)
        local = order[self.rank:: self.num_replicas]
        batches = np.split(local, range(self.micro_batch_size, len(local), self.micro_batch_size), axis=0)
        if len(batches) != self.local_num_batches:
            raise RuntimeError(f"Number of calculated indices {len(batches)} is not equal to calculated number of local "
                               f"batches {self.local_num_batches}.")
        return batches

    def __iter__(self) -> Iterator[List[int]]:
        batches = self._make_batches()
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch + 1)
            visit = torch.randperm(self.local_num_batches, generator=g)
        else:
            visit = torch.arange(0, self.local_num_batches)
        for i in visit.tolist():
            yield batches[i].tolist()

    def __len__(self) -> int:
        return self.local_num_batches

    def padding_fraction(self) -> float:
        """share of padded samples over one epoch of this rank's batches (a shaping diagnostic, not in the reference)"""
        pad = tot = 0.0
        for b in self._make_batches():
            d = self.durations[b]
            tot += float(d.max()) * len(d)
            pad += float((d.max() - d).sum())
        return pad / tot if tot else 0.0
