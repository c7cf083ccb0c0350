"""Input side of the hot path on CPU: the manifest reader, the character parser, the padding
collate function and the semi-sorted batch sampler are pinned against outputs of the reference's own code
(tests/golden/ref_data_pipeline.json, written by oracle/make_golden.py::make_data_fixture in the build container); the
datasets, the duration shaping and the staged loader are then exercised end to end on a synthetic WAV corpus."""
import json
import os
import wave

import numpy as np
import pytest
import torch

from nemo_amd.data import (AudioToBPEDataset, AudioToCharDataset, CharParser, DeviceBatchLoader, SemiSortBatchSampler,
                           _speech_collate_fn, item_iter, load_audio)

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_data_pipeline.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


def test_semi_sorted_sampler_matches_the_reference_class(gold):
    durations = gold["durations"]
    for case in gold["sampler"]:
        for epoch, ranks in enumerate(case["epochs"]):
            for rank, want in enumerate(ranks):
                sm = SemiSortBatchSampler(rank, case["world"], durations, case["batch_size"], case["shuffle"],
                                          case["drop_last"], case["randomization_factor"], case["seed"])
                sm.set_epoch(epoch)
                np.random.seed(case["np_seed"] + epoch)
                got = [list(b) for b in sm]
                assert got == want, (case["world"], epoch, rank)
                assert len(sm) == len(want)


def test_semi_sorted_sampler_shapes_batches_and_keeps_ranks_in_step():
    rs = np.random.RandomState(0)
    durations = rs.uniform(5.0, 30.0, size=2000).astype(np.float32)  # SURVEY.md cfg 5: uniform 5..30 s
    world, bs = 8, 32
    samplers = [SemiSortBatchSampler(r, world, durations, bs, seed=11, synced_rng=True) for r in range(world)]
    per_rank = [list(s) for s in samplers]
    # every rank draws the same ordering without any help from a global seed: the epoch is partitioned
    seen = np.concatenate([np.concatenate(b) for b in per_rank])
    assert len(set(seen.tolist())) == len(durations)            # every utterance appears
    assert len(seen) - len(durations) < world * bs              # only the padding batches repeat utterances
    assert len({len(b) for b in per_rank}) == 1                 # same number of steps on every rank
    # the k-th batch of every rank covers the same stretch of durations: padded lengths agree to a few percent
    longest = np.array([[durations[b].max() for b in batches] for batches in per_rank])  # [rank, step]
    spread = (longest.max(0) - longest.min(0)) / longest.max(0)
    assert np.median(spread) < 0.06 and spread.max() < 0.25
    # a batch spans world*bs neighbours of the ordering: ~12 % padding here against ~40 % for random batches
    assert samplers[0].padding_fraction() < 0.15
    rnd = rs.permutation(len(durations))[: (len(durations) // bs) * bs].reshape(-1, bs)
    random_pad = float(sum((durations[b].max() - durations[b]).sum() for b in rnd) / sum(durations[b].max() * bs for b in rnd))
    assert random_pad > 0.3
    # a new epoch reshuffles
    samplers[0].set_epoch(1)
    assert [list(b) for b in samplers[0]] != per_rank[0]


def test_collate_matches_the_reference_function(gold):
    for case in gold["collate"]:
        batch = []
        for i, (sig, tok) in enumerate(zip(case["signals"], case["tokens"])):
            item = (torch.tensor(sig, dtype=torch.float32), torch.tensor(len(sig)).long(), torch.tensor(tok).long(),
                    torch.tensor(len(tok)).long())
            batch.append(item + (100 + i,) if case["with_ids"] else item)
        out = _speech_collate_fn(batch, case["pad_id"])
        assert len(out) == len(case["out"])
        for got, want, dt in zip(out, case["out"], case["out_dtypes"]):
            assert str(got.dtype) == dt
            assert torch.equal(got, torch.tensor(want, dtype=got.dtype))
    with pytest.raises(ValueError):
        _speech_collate_fn([(torch.zeros(3), torch.tensor(3), torch.zeros(1).long())], 0)


def test_char_parser_matches_the_reference_class(gold):
    for case in gold["parser"]:
        p = CharParser(case["labels"], **case["kwargs"])
        assert [p(t) for t in case["texts"]] == case["ids"]


def test_manifest_reader_matches_the_reference(gold, tmp_path):
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "wavs"))
    open(os.path.join(d, "wavs", "a.wav"), "wb").close()
    with open(os.path.join(d, "t.txt"), "w") as f:
        f.write("from a\nfile\n")
    mpath = os.path.join(d, "m.json")
    with open(mpath, "w") as f:
        for ln in gold["manifest"]["lines"]:
            f.write(ln.replace("<DIR>", d) + "\n")
        f.write("\n")
    items = list(item_iter(mpath))
    assert len(items) == len(gold["manifest"]["items"])
    for got, want in zip(items, gold["manifest"]["items"]):
        for k, v in want.items():
            g = got[k].replace(d, "<DIR>") if k == "audio_file" else got[k]
            assert g == v, (k, g, v)
    with open(mpath, "a") as f:
        f.write("{not json\n")
    with pytest.raises(RuntimeError):
        list(item_iter(mpath))


def _write_wav(path, x, sr=16000, width=2, channels=1):
    with wave.open(path, "wb") as f:
        f.setnchannels(channels); f.setsampwidth(width); f.setframerate(sr)
        f.writeframes(np.asarray(x, dtype="<i2").tobytes())


def _corpus(tmp_path, n=23, sr=16000, seed=0):
    rs = np.random.RandomState(seed)
    words = ["ab", "cab", "bed", "ace", "dad", "a"]
    lines, pcm = [], []
    for i in range(n):
        dur = float(rs.randint(2000, 9000)) / sr
        x = rs.randint(-3000, 3000, size=int(dur * sr)).astype(np.int16)
        _write_wav(str(tmp_path / f"u{i}.wav"), x, sr)
        pcm.append(x)
        text = " ".join(rs.choice(words, size=rs.randint(1, 4)))
        lines.append(dict(audio_filepath=f"u{i}.wav", duration=dur, text=text))
    m = str(tmp_path / "train.json")
    with open(m, "w") as f:
        for ln in lines:
            f.write(json.dumps(ln) + "\n")
    return m, lines, pcm


def test_load_audio_scaling_offset_duration_and_channels(tmp_path):
    x = (np.arange(-800, 800) * 40).astype(np.int16)
    p = str(tmp_path / "a.wav")
    _write_wav(p, x)
    y = load_audio(p, 16000)
    assert y.dtype == torch.float32 and torch.equal(y, torch.from_numpy(x.astype(np.float32) / 32768.0))
    z = load_audio(p, 16000, offset=0.01, duration=0.05)  # seek(int(0.01*sr)) / read(int(0.05*sr))
    assert torch.equal(z, y[160:160 + 800])
    assert torch.equal(load_audio(p, 16000, int_values=True), y)  # int32 read scaled by 2^-31 is the same number
    st = np.stack([x, -x // 2], axis=1)
    ps = str(tmp_path / "s.wav")
    _write_wav(ps, st.reshape(-1), channels=2)
    assert torch.allclose(load_audio(ps, 16000, channel_selector="average"),
                          torch.from_numpy(st.astype(np.float32).mean(1) / 32768.0), atol=1e-7)
    assert torch.equal(load_audio(ps, 16000, channel_selector=1), torch.from_numpy((-x // 2).astype(np.float32) / 32768.0))
    with pytest.raises(ValueError):
        load_audio(ps, 16000)
    with pytest.raises(ValueError):
        load_audio(p, 8000)


def test_char_dataset_end_to_end_with_semi_sorted_batches(tmp_path):
    m, lines, pcm = _corpus(tmp_path)
    labels = [" ", "a", "b", "c", "d", "e"]
    ds = AudioToCharDataset(m, labels=labels, sample_rate=16000, max_duration=0.5, min_duration=0.13, return_sample_id=True)
    kept = [i for i, ln in enumerate(lines) if 0.13 <= ln["duration"] <= 0.5]
    assert len(ds) == len(kept) and len(ds) < len(lines)
    f, fl, t, tl, idx = ds[0]
    src = kept[0]
    assert torch.equal(f, torch.from_numpy(pcm[src].astype(np.float32) / 32768.0)) and int(fl) == len(pcm[src])
    assert t.tolist() == [labels.index(c) for c in lines[src]["text"]] and int(tl) == len(lines[src]["text"])
    sm = SemiSortBatchSampler(0, 1, ds.durations, batch_size=4, seed=3, synced_rng=True)
    dl = torch.utils.data.DataLoader(ds, batch_size=None, sampler=sm, collate_fn=ds._collate_fn)
    n = 0
    for sig, sl, tok, tkl, ids in dl:
        assert sig.shape == (len(ids), int(sl.max())) and tok.shape == (len(ids), int(tkl.max()))
        for r, i in enumerate(ids.tolist()):
            assert torch.equal(sig[r, : int(sl[r])], ds[i][0]) and float(sig[r, int(sl[r]):].abs().sum()) == 0.0
        n += len(ids)
    assert n == len(ds)


def test_bpe_dataset_token_rules(tmp_path):
    m, lines, _ = _corpus(tmp_path, n=5)

    class Tok:  # the interface AudioToBPEDataset needs (audio_to_text.py:669-700)
        bos_id, eos_id, pad_id = 1, 2, 3

        def text_to_ids(self, text):
            return [10 + len(w) for w in text.split()]
    ds = AudioToBPEDataset(m, tokenizer=Tok(), sample_rate=16000)
    _, _, t, tl = ds[2]
    assert t.tolist() == [1] + [10 + len(w) for w in lines[2]["text"].split()] + [2] and int(tl) == len(t)
    assert ds.manifest_processor.pad_id == 3
    Tok.bos_id, Tok.eos_id, Tok.pad_id = 0, -1, 0  # ids <= 0 are "absent": no wrapping, pad 0
    ds = AudioToBPEDataset(m, tokenizer=Tok(), sample_rate=16000)
    assert ds[2][2].tolist() == [10 + len(w) for w in lines[2]["text"].split()] and ds.manifest_processor.pad_id == 0
    batch = ds._collate_fn([ds[i] for i in range(5)])
    assert batch[0].shape[0] == 5 and batch[2].dtype == torch.int64


def test_device_batch_loader_passes_batches_through_on_cpu_and_surfaces_errors():
    batches = [(torch.full((2, 5), float(i)), torch.tensor([5, 3]), torch.ones(2, 2).long(), torch.tensor([2, 1]))
               for i in range(7)]
    got = list(DeviceBatchLoader(batches, "cpu", prefetch=2))
    assert len(got) == 7 and all(torch.equal(a[0], b[0]) for a, b in zip(got, batches))

    def bad():
        yield batches[0]
        raise OSError("disk gone")
    it = iter(DeviceBatchLoader(bad(), "cpu"))
    next(it)
    with pytest.raises(OSError):
        next(it)
    # leaving the loop early does not leave the worker blocked on a full queue
    for k, _ in enumerate(DeviceBatchLoader(batches, "cpu", prefetch=1)):
        if k == 1:
            break


def test_semi_sorted_sampler_invariants_for_arbitrary_sizes():
    """property test over corpus size / world size / batch size / drop_last: every rank takes the same number of steps (the
    all-reduce would dead-lock otherwise), only valid indices appear, nothing is lost beyond the dropped tail, and repeats
    are bounded by the padding batches"""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=120, deadline=None)
    @given(n=st.integers(1, 300), world=st.sampled_from([1, 2, 3, 4, 8]), bs=st.integers(1, 16), drop_last=st.booleans(),
           seed=st.integers(0, 10_000))
    def check(n, world, bs, drop_last, seed):
        rs = np.random.RandomState(seed)
        durations = rs.uniform(0.5, 30.0, size=n)
        per_rank = []
        for r in range(world):
            sm = SemiSortBatchSampler(r, world, durations, bs, batch_shuffle=True, drop_last=drop_last, seed=seed,
                                      synced_rng=True)
            batches = list(sm)
            assert len(batches) == len(sm)
            per_rank.append(batches)
        assert len({len(b) for b in per_rank}) == 1
        flat = [i for b in per_rank for x in b for i in x]
        assert all(0 <= i < n for i in flat)
        kept = n - (n % bs if drop_last else 0)
        if kept == 0:
            assert flat == []
            return
        assert len(set(flat)) >= kept - 0 if not drop_last else len(set(flat)) <= n
        if not drop_last:
            assert set(flat) == set(range(n))
        assert len(flat) - len(set(flat)) <= world * bs  # repeats come from the padding batches only
        assert all(len(x) <= bs for b in per_rank for x in b)
    check()


def test_duration_bucket_sampler_follows_the_reference_bucketing_semantics():
    """static bucketing (convert_to_tarred_audio_dataset.py:749-757 equal-width duration ranges; BucketingDataset batches inside one
    bucket; RandomizedChainDataset(rnd_seed=0) visits the buckets in RandomState(0)'s permutation on EVERY rank, audio_to_text.py:
    1374-1389; calc_bucketing_batch_sizes' linear scaling, audio_to_text_dataset.py:961-1000)"""
    import numpy as np
    from nemo_amd.data import DurationBucketBatchSampler
    rng = np.random.RandomState(3)
    durs = rng.uniform(5.0, 30.0, size=2000)
    world, nb = 4, 5
    samplers = [DurationBucketBatchSampler(r, world, durs, 16, nb, min_duration=5.0, max_duration=30.0) for r in range(world)]
    per_rank = [list(s) for s in samplers]
    assert len({len(b) for b in per_rank}) == 1 and all(len(b) == len(s) for b, s in zip(per_rank, samplers))
    seen = [i for b in per_rank for batch in b for i in batch]
    assert len(seen) == len(set(seen))                                  # nothing is read twice in an epoch
    width = 25.0 / nb
    visit = []
    for step in range(len(per_rank[0])):
        buckets = {int((durs[i] - 5.0) // width) for r in range(world) for i in per_rank[r][step]}
        assert len(buckets) == 1, (step, buckets)                       # every rank is inside the same bucket at the same step
        b = buckets.pop()
        if not visit or visit[-1] != b:
            visit.append(b)
    assert visit == np.random.RandomState(0).permutation(nb).tolist()  # RandomizedChainDataset(rnd_seed=0)
    # padded share: a batch never spans more than one bucket width
    assert samplers[0].padding_fraction() < width / 2 / 5.0
    # adaptive batch sizes: (buckets_num - idx) * bucketing_batch_size, batch_size must be 1
    s = DurationBucketBatchSampler(0, 1, durs, 1, nb, 5.0, 30.0, bucketing_strategy="fixed_order", bucketing_batch_size=8)
    sizes = [len(b) for b in s]
    assert s.batch_sizes == [40, 32, 24, 16, 8] and sizes[0] == 40 and sizes[-2] in (8,)
    order = [s.bucket_of(durs[b[0]]) for b in s]
    assert order == sorted(order)                                       # fixed_order: bucket 0 first
    with pytest.raises(ValueError, match="batch_size should be set to one"):
        DurationBucketBatchSampler(0, 1, durs, 4, nb, 5.0, 30.0, bucketing_batch_size=8)
    with pytest.raises(ValueError, match="is not supported"):
        DurationBucketBatchSampler(0, 1, durs, 4, nb, bucketing_strategy="sorted")


@pytest.mark.parametrize("world", [2, 4, 8])
def test_batch_samplers_give_every_rank_the_same_number_of_batches(world):
    """what keeps a data-parallel epoch from dead-locking: every rank must run the SAME number of steps (each step holds collectives),
    whatever the data-set size; and the ranks' shards must tile the data set -- SemiSortBatchSampler with the synced generator covers
    every utterance (a few repeated to even out the ranks, like torch's DistributedSampler), the duration-bucket sampler hands out
    no utterance twice.  (asr_batching.py:27-240; audio_to_text_dataset.py:961-1000)"""
    from nemo_amd.data import DurationBucketBatchSampler, SemiSortBatchSampler
    rng = np.random.default_rng(world)
    for n in (97, 640, 1001):
        dur = rng.uniform(5, 30, size=n).tolist()
        for drop_last in (False, True):
            per_rank = []
            for r in range(world):
                s = SemiSortBatchSampler(global_rank=r, world_size=world, durations=dur, batch_size=4, batch_shuffle=True,
                                         drop_last=drop_last, randomization_factor=0.1, seed=42, synced_rng=True)
                b = list(iter(s))
                assert len(b) == len(s)
                per_rank.append(b)
            assert len({len(b) for b in per_rank}) == 1, (n, drop_last, [len(b) for b in per_rank])
            flat = [i for b in per_rank for x in b for i in x]
            assert len(set(flat)) >= n - (4 * world if drop_last else 0)          # every utterance (minus at most one dropped round)
            assert len(flat) - len(set(flat)) <= 4 * world                         # repeats only to even out the last round
            per_rank = []
            for r in range(world):
                s = DurationBucketBatchSampler(global_rank=r, world_size=world, durations=dur, batch_size=4, buckets_num=4, seed=1,
                                               drop_last=drop_last)
                per_rank.append(list(iter(s)))
            assert len({len(b) for b in per_rank}) == 1
            flat = [i for b in per_rank for x in b for i in x]
            assert len(flat) == len(set(flat))
