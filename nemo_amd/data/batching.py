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


class DurationBucketBatchSampler:
    """Static duration bucketing, as the reference trains its Conformer recipes (`bucketing_strategy: synced_randomized`,
    examples/asr/conf/conformer/conformer_ctc_bpe.yaml:66): the corpus is split OFF-LINE into `buckets_num` datasets over
    equal-WIDTH duration ranges of [min_duration, max_duration) (scripts/speech_recognition/convert_to_tarred_audio_dataset.py:
    749-757), every bucket is batched on its own (`BucketingDataset`, nemo/collections/asr/data/audio_to_text.py:1322-1371; fixed
    batch size, or `bucketing_batch_size` scaled by (buckets_num - bucket index): shorter utterances, larger batches --
    audio_to_text_dataset.py:961-1000 `calc_bucketing_batch_sizes`), each rank reads its own shard of every bucket, and the
    buckets are visited one after the other in a permutation drawn from `np.random.RandomState(rnd_seed)`
    (`RandomizedChainDataset`, audio_to_text.py:1374-1389): seed 0 on every rank for 'synced_randomized' (all ranks are inside
    the SAME bucket at the same step, i.e. pad to similar lengths -- what the gradient all-reduce's slowest rank needs),
    a per-rank random seed for 'fully_randomized', the given order for 'fixed_order' (audio_to_text_dataset.py:948-958).

    This is the map-style restatement for in-memory manifests: it yields lists of utterance indices.  Inside a bucket the
    utterances are shuffled (the tarred loader's shuffle buffer) with `seed + epoch`, dealt to the ranks round-robin and cut
    into batches; every rank gets the same number of batches per bucket (the surplus of the dealing is dropped, as the tarred
    datasets' equal-length sharding does)."""

    def __init__(self, global_rank: int, world_size: int, durations: Sequence[float], batch_size: int, buckets_num: int,
                 min_duration: Optional[float] = None, max_duration: Optional[float] = None,
                 bucketing_strategy: str = "synced_randomized", bucketing_batch_size=None, seed: int = 0,
                 drop_last: bool = False) -> None:
        if bucketing_strategy not in ("fixed_order", "synced_randomized", "fully_randomized"):
            raise ValueError(f"bucketing_strategy={bucketing_strategy} is not supported! Supported strategies are "
                             "[fixed_order, fully_randomized, synced_randomized].")  # audio_to_text_dataset.py:955-958
        if buckets_num < 1:
            raise ValueError("buckets_num must be >= 1")
        self.rank, self.world = global_rank, world_size
        self.durations = np.asarray(durations, dtype=np.float64)
        self.buckets_num, self.strategy, self.seed, self.drop_last, self.epoch = buckets_num, bucketing_strategy, seed, drop_last, 0
        lo = float(self.durations.min()) if min_duration is None else float(min_duration)
        hi = float(self.durations.max()) if max_duration is None else float(max_duration)
        width = (hi - lo) / float(buckets_num)
        self.edges = [lo + i * width for i in range(buckets_num)] + [hi + 1e-5]  # (the last bucket includes max_duration)
        if bucketing_batch_size is None:
            self.batch_sizes = [batch_size] * buckets_num
        elif isinstance(bucketing_batch_size, int):  # linear scaling, calc_bucketing_batch_sizes
            if batch_size != 1:
                raise ValueError("batch_size should be set to one when bucketing_batch_size is set and adaptive bucketing is "
                                 f"enabled (batch_size={batch_size}!")
            self.batch_sizes = [(buckets_num - i) * bucketing_batch_size for i in range(buckets_num)]
        else:
            self.batch_sizes = list(bucketing_batch_size)
            if len(self.batch_sizes) != buckets_num:
                raise ValueError(f"batch_size should have the same length as the number of buckets ({len(self.batch_sizes)}!="
                                 f"{buckets_num}) ")
        # rnd_seed of RandomizedChainDataset: 0 / random.randint(0, 30000) + rank; the generator lives as long as the dataset, so
        # successive epochs draw successive permutations from the same stream
        if bucketing_strategy == "fully_randomized":
            import random
            self._chain_rng = np.random.RandomState(random.randint(0, 30000) + global_rank)
        else:
            self._chain_rng = np.random.RandomState(0)

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def bucket_of(self, duration: float) -> int:
        for i in range(self.buckets_num):
            if self.edges[i] <= duration < self.edges[i + 1]:
                return i
        return -1  # outside [min_duration, max_duration]: filtered out, as the bucket manifests do

    def _bucket_batches(self, b: int) -> List[List[int]]:
        idx = np.nonzero((self.durations >= self.edges[b]) & (self.durations < self.edges[b + 1]))[0]
        rng = np.random.RandomState(self.seed + self.epoch * 1000 + b)
        idx = idx[rng.permutation(len(idx))]
        per_rank = len(idx) // self.world
        mine = idx[self.rank: per_rank * self.world: self.world]
        bs = self.batch_sizes[b]
        n_full = len(mine) // bs
        out = [mine[i * bs:(i + 1) * bs].tolist() for i in range(n_full)]
        if not self.drop_last and len(mine) % bs:
            out.append(mine[n_full * bs:].tolist())
        return out

    def __iter__(self) -> Iterator[List[int]]:
        order = list(range(self.buckets_num)) if self.strategy == "fixed_order" else self._chain_rng.permutation(self.buckets_num).tolist()
        for b in order:
            yield from self._bucket_batches(b)

    def __len__(self) -> int:
        n = 0
        for b in range(self.buckets_num):
            cnt = int(((self.durations >= self.edges[b]) & (self.durations < self.edges[b + 1])).sum()) // self.world
            n += cnt // self.batch_sizes[b] + (0 if self.drop_last or cnt % self.batch_sizes[b] == 0 else 1)
        return n

    def padding_fraction(self) -> float:
        pad = tot = 0.0
        for b in range(self.buckets_num):
            for batch in self._bucket_batches(b):
                d = self.durations[batch]
                tot += float(d.max()) * len(d)
                pad += float((d.max() - d).sum())
        return pad / tot if tot else 0.0
