#!/usr/bin/env python
"""Headline benchmark: audio-sec/s training throughput, Conformer-CTC-Large (BASELINE.json), MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one full optimizer step of the drop-in EncDecCTCModel on a synthetic batch resident in HBM:
log-mel front-end -> conv sub-sampling -> 18 Conformer blocks -> decoder -> CTC loss -> full backward -> bucketed RCCL
gradient all-reduce (N > 1, overlapped on a side stream) -> fused AdamW + Noam step; train mode (dropout, dither,
batch-statistics BatchNorm with SyncBN across ranks), bf16 compute with fp32 master weights.  Per-GPU batch is fixed
(weak scaling): 32 x 20 s clips of 16 kHz audio, 60 target tokens each, vocab 128 + blank.
Prints ONE JSON line (rank 0) -- see the README/DESIGN.md for the fields; `roofline` is measured live with HIP events
around every launch of the dominant GEMM kernel in one extra step, `cpu_baseline` times the CPU oracle (oracle/) on the
host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", default="large", choices=["xs", "small", "sm", "medium", "ml", "large"])
    ap.add_argument("--model", default="ctc", choices=["ctc", "transducer", "squeezeformer"],
                    help="ctc = Conformer-CTC (BASELINE configs[1]/[2], the headline); transducer = FastConformer-Transducer "
                         "(configs[3]: x8 dw_striding encoder, LSTM prediction network, fused joint + RNN-T loss); squeezeformer = "
                         "Squeezeformer-CTC (configs[4]: --size medium, SpecAugment on)")
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--secs", type=float, default=20.0)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-spec-augment", action="store_true", help="drop the recipe's SpecAugment (on by default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--var-len", default=None, metavar="LO:HI",
                    help="variable-length workload (BASELINE configs[4]: '5:30'): utterance durations uniform in [LO, HI] seconds, "
                         "batches shaped by --sampler, value = VALID audio seconds per second")
    ap.add_argument("--sampler", default="semisort", choices=["semisort", "bucket", "random"],
                    help="--var-len batch shaping: semisort = SemiSortBatchSampler (asr_batching.py:27-204), bucket = static "
                         "duration buckets (BucketingDataset / synced_randomized), random = unshaped batches (padding stress)")
    ap.add_argument("--buckets", type=int, default=8)
    ap.add_argument("--packed", default=None, choices=["auto", "0", "1"],
                    help="--var-len: packed token chain (SURVEY 8 f1; encoder.packed_rows): auto = the encoder's default (pack when the "
                         "host knows the lengths and >= 2 %% of the frames are padding), 0 = padded rows, 1 = always")
    ap.add_argument("--pad-to", type=int, default=None, metavar="N",
                    help="--var-len: the featurizer's own `pad_to` option (features.py:501; recipes ship 0 or 16): the feature frames of a "
                         "batch are padded to a multiple of N, so a duration-shaped loader produces a few dozen padded lengths instead of "
                         "one per batch -- each length's launch sequence can then be recorded once and replayed (--launch)")
    ap.add_argument("--launch", default=None, choices=["auto", "live", "tape"],
                    help="encoder launch mode (default: MI355X_GRAPHS or auto): live = every kernel from the Python sequencer, tape = "
                         "recorded launch sequences replayed per padded length, auto = both timed on the device, the faster kept")
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--cpu-steps", type=int, default=2)
    return ap.parse_args()


def synthetic_batch(B, secs, vocab=128, seed=1234):
    """SURVEY.md section 8(d) synthetic inputs: audio 0.1*randn (the reference's OOMptimizer batch construction), tokens
    randint(0, vocab), U = 3*secs, full lengths.  (Restated here: the measured leg does not touch oracle/.)"""
    g = torch.Generator().manual_seed(seed)
    S = int(round(16000 * secs))
    audio = 0.1 * torch.randn(B, S, generator=g)
    U = max(1, int(3 * secs))
    tokens = torch.randint(0, vocab, (B, U), generator=g)
    return audio, torch.full((B,), S, dtype=torch.int64), tokens, torch.full((B,), U, dtype=torch.int64)


def var_len_batches(a, rank, world, vocab, dev):
    """--var-len LO:HI: one synthetic corpus of (warmup + steps) x batch x world utterances with durations uniform in [LO, HI] s
    (the same table on every rank), cut into batches by the chosen sampler, every batch padded to ITS longest utterance
    (_speech_collate_fn semantics) and resident in HBM before the clock starts.  Returns (batches, valid seconds per batch)."""
    import numpy as np
    from nemo_amd.data import DurationBucketBatchSampler, SemiSortBatchSampler
    lo, hi = (float(x) for x in a.var_len.split(":"))
    n_steps = a.warmup + a.steps
    n = n_steps * a.batch * world
    if a.sampler == "bucket":  # only full batches are kept per bucket and rank: a corpus with room for the dropped tails
        n += a.buckets * a.batch * world
    durs = np.random.RandomState(4321).uniform(lo, hi, size=n)
    if a.sampler == "semisort":
        sm = SemiSortBatchSampler(rank, world, durs, a.batch, batch_shuffle=True, drop_last=True, randomization_factor=0.1, seed=42,
                                  synced_rng=True)
    elif a.sampler == "bucket":
        sm = DurationBucketBatchSampler(rank, world, durs, a.batch, a.buckets, min_duration=lo, max_duration=hi, drop_last=True)
    else:
        perm = np.random.RandomState(7).permutation(n)[rank::world]
        sm = [perm[i * a.batch:(i + 1) * a.batch].tolist() for i in range(len(perm) // a.batch)]
    idx_batches = list(sm)
    if not idx_batches:
        raise SystemExit("bench.py --var-len: the corpus is too small for one full batch per rank")
    if len(idx_batches) < n_steps:  # (bucket tails dropped): cycle -- the shapes repeat, the work per step is unchanged
        idx_batches = (idx_batches * (n_steps // max(1, len(idx_batches)) + 1))
    idx_batches = idx_batches[:n_steps]
    if a.warmup > 0:
        # the un-timed steps start with the LONGEST batch of the run: the caching allocator and the workspaces reach their final
        # sizes there (what a long training run reaches after its first pass over the longest bucket), instead of growing --
        # hipMalloc by hipMalloc -- inside the 20 timed steps
        # (a COPY of it: the timed steps keep the sampled sequence, worst-padded batch included -- moving it out of the timed window
        #  biased the valid-audio rate upwards, round-3 advisor finding; the last warm-up batch is dropped to keep the step count)
        longest = max(range(len(idx_batches)), key=lambda i: float(durs[idx_batches[i]].max()))
        idx_batches = [idx_batches[longest]] + idx_batches[1:a.warmup] + idx_batches[a.warmup:]
    g = torch.Generator().manual_seed(1234 + rank)
    batches, valid = [], []
    for ib in idx_batches:
        d = durs[ib]
        lens = torch.tensor(np.round(d * 16000).astype(np.int64))
        S = int(lens.max())
        audio = 0.1 * torch.randn(len(ib), S, generator=g)
        audio *= (torch.arange(S).unsqueeze(0) < lens.unsqueeze(1))          # collate pads with zeros
        tl = torch.tensor(np.maximum(1, (3 * d).astype(np.int64)))
        tok = torch.randint(0, vocab, (len(ib), int(tl.max())), generator=g)
        lens_dev = lens.to(dev)
        lens_dev.host_lengths = lens   # what the input pipeline knows anyway (nemo_amd/data/loader.py attaches it the same way):
        # lets the encoder size a PACKED launch sequence -- the valid frames only -- without reading the lengths back
        batches.append([audio.to(dev), lens_dev, tok.to(dev), tl.to(dev)])
        valid.append(float(d.sum()))
    pad = 1.0 - sum(valid) / sum(float(durs[ib].max()) * len(ib) for ib in idx_batches)
    return batches, valid, {"durations": f"uniform {lo:g}-{hi:g} s", "sampler": a.sampler + (f" ({a.buckets} buckets)" if a.sampler == "bucket" else ""),
                            "padded_sample_fraction": round(pad, 4), "distinct_padded_lengths": len({b[0].shape[1] for b in batches})}


def cpu_baseline(size, secs, vocab, batch, steps, budget_s=40.0):
    """The CPU leg: forward + backward + AdamW of Conformer-CTC in fp32, train mode, on the host cores.
    kind "reference": the reference's OWN modules (FilterbankFeatures + ConformerEncoder loaded verbatim from /root/reference through
    oracle/ref_shim.py, decoder / CTCLoss as the 3-line restatements of SURVEY.md section 8c) -- only where that tree exists (the build
    container); kind "port": the oracle restatement (oracle/conformer_ref.py, pinned to the reference by tests/test_oracle_pinning.py)
    -- the GPU box, where /root/reference does not exist.  (threads, batch) are SWEPT (over-subscribing the host costs more than
    half: 128 threads on B = 2 measured 2.3 audio-s/s where 16 give 21): one probe step per point inside the budget, every batch
    size gets at least one probe, then the median of >= 3 steps at the best point."""
    import os as _os
    from oracle import conformer_ref as R
    from oracle import ref_shim
    cfg = getattr(R.ConformerCfg, size)(vocab=vocab)
    ncpu = _os.cpu_count() or 1
    t_start = time.perf_counter()
    use_ref = ref_shim.reference_available() and _os.environ.get("BENCH_CPU_KIND", "auto") != "port"

    def make(batch_):
        data = R.synthetic_batch(batch_, secs, vocab=vocab, seed=1234)
        if use_ref:
            torch.manual_seed(0)
            m = ref_shim.ReferenceCTCModel(cfg.d_model, cfg.n_heads, cfg.n_layers, vocab=vocab, dropout=0.1, dropout_att=0.1,
                                           dither=1e-5).train()
            opt = torch.optim.AdamW(m.parameters(), lr=1e-4, betas=(0.9, 0.98), weight_decay=1e-3)
            return m, opt, data
        P = R.init_params(cfg, seed=0, nonzero_pos_bias=False)
        keys = R.trainable_keys(P)
        for k in keys:
            P[k].requires_grad_(True)
        opt = torch.optim.AdamW([P[k] for k in keys], lr=1e-4, betas=(0.9, 0.98), weight_decay=1e-3)
        return P, opt, data

    def step(P, opt, data):
        audio, alen, tok, tl = data
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        if use_ref:
            loss = P(audio, alen, tok, tl)[0]
        else:
            noise = torch.randn_like(audio)
            loss = R.model_forward(P, cfg, audio, alen, tok, tl, train=True, noise=noise, dither=1e-5)["loss"]
        loss.backward()
        opt.step()
        return time.perf_counter() - t0

    threads = [t for t in (16, 32, 64) if t <= ncpu] or [ncpu]
    batches = sorted({batch, max(batch, 8)})
    probes, states = {}, {}
    # (the reference's logger writes INFO lines to stdout: this process's stdout carries exactly one JSON line, so the leg runs
    # with fd 1 pointed at stderr)
    sys.stdout.flush()
    saved_fd = _os.dup(1)
    _os.dup2(2, 1)
    try:
        best, times = _cpu_sweep(threads, batches, make, step, secs, budget_s, t_start, steps, probes, states)
    finally:
        sys.stdout.flush()
        _os.dup2(saved_fd, 1)
        _os.close(saved_fd)
    th, b_ = best
    times.sort()
    med = times[len(times) // 2]
    # how the port compares with the reference's own modules on one host (measured in the build container, where both exist:
    # tools/cpu_calibration.py -> profiles/r4_cpu_calibration.json): > 1 means the port is the FASTER of the two, i.e. the
    # GPU / CPU ratio of this line is conservative
    calib = None
    try:
        with open(_os.path.join(ROOT, "profiles", "r4_cpu_calibration.json")) as f:
            cj = json.load(f)
        calib = {"port_over_reference": cj["port_over_reference"], "threads": cj["threads"], "source": "profiles/r4_cpu_calibration.json "
                 "(tools/cpu_calibration.py, build container: reference modules through oracle/ref_shim.py vs oracle/conformer_ref.py, "
                 "alternating legs)"}
    except (OSError, ValueError, KeyError):
        pass
    return {"value": round(b_ * secs / med, 2), "unit": "audio-sec/s", "cores": th, "threads": th, "host_cpus": ncpu,
            "kind": "reference" if use_ref else "port", "calibration": calib,
            "sample": f"Conformer-CTC-{size} fp32 train step (fwd+bwd+AdamW), B={b_}x{secs:g}s, median of {len(times)} steps at the best "
                      f"of {len(probes)} (threads, batch) probes; "
                      + ("the reference's own FilterbankFeatures + ConformerEncoder through oracle/ref_shim.py" if use_ref else
                         "oracle/conformer_ref.py (the reference tree does not exist on this box)"),
            "cpu_leg_seconds": round(time.perf_counter() - t_start, 1),
            "sweep_audio_sec_per_s": {f"threads{t}_b{b}": round(v, 2) for (t, b), v in sorted(probes.items())}}


def _cpu_sweep(threads, batches, make, step, secs, budget_s, t_start, steps, probes, states):
    for bi, b_ in enumerate(batches):
        states[b_] = make(b_)
        torch.set_num_threads(threads[min(1, len(threads) - 1)])
        step(*states[b_])  # warm-up (allocator, thread pool)
        share = budget_s * (bi + 1) / len(batches) * 0.7
        for ti, th in enumerate(threads):
            if ti > 0 and time.perf_counter() - t_start > share:
                break
            torch.set_num_threads(th)
            probes[(th, b_)] = b_ * secs / step(*states[b_])
    (th, b_), _ = max(probes.items(), key=lambda kv: kv[1])
    torch.set_num_threads(th)
    times = []
    for _ in range(max(3, steps)):
        times.append(step(*states[b_]))
        if time.perf_counter() - t_start > budget_s * 1.5 and len(times) >= 3:
            break
    return (th, b_), times


def _source_hash():
    """sha256 over the GEMM kernel sources: off-line PMC figures are only reported for the code they were measured on"""
    import hashlib
    h = hashlib.sha256()
    for f in ("gemm.hip", "common.h"):
        with open(os.path.join(ROOT, "nemo_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def vendor_ceiling(dev):
    """Calibration, not a leg of the product path: what the vendor library (hipBLASLt through torch.matmul) reaches on THIS box
    with a plain store epilogue, next to mi355x_gemm on the same operands, for the three shapes VERDICT r4 item 1 names --
    WARM (one operand set re-used by every launch: inputs and the 16-66 MB output stay in the 256 MiB Infinity Cache) and COLD
    (`rot` independent operand / output sets in turn, > 288 MiB between two uses of a set: where a launch finds its operands inside
    the training step).  The two regimes order the kernels differently (profiles/r5_vendor_gemm_anatomy.md): the top-level keys
    keep round 4's warm numbers on the FFN1 shape, `shapes` carries both regimes for all three."""
    from nemo_amd import ops

    def t(fs, iters=24):
        n = len(fs)
        for i in range(max(3, n)):
            fs[i % n]()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fs[i % n]()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e-3

    out = {"epilogue": "plain bf16 store", "shapes": []}
    try:
        for (M, N, K) in ((16032, 2048, 512), (16032, 512, 2048), (16032, 512, 512)):
            rot = max(2, int(320e6 // (2 * (M * K + N * K + M * N))) + 1)
            sets = [(torch.randn(M, K, device=dev).to(torch.bfloat16), torch.randn(N, K, device=dev).to(torch.bfloat16),
                     torch.empty(M, N, device=dev, dtype=torch.bfloat16)) for _ in range(rot)]
            lib = lambda A, B, C: (lambda: torch.matmul(A, B.t(), out=C))
            own = lambda A, B, C: (lambda: ops.gemm(A, B, C, M, N, K, K, K, N))
            fl = 2.0 * M * N * K
            row = {"shape": [M, N, K], "operand_sets_cold": rot}
            for who, mk in (("hipblaslt", lib), ("mi355x_gemm", own)):
                tw, tc = t([mk(*sets[0])]), t([mk(*s_) for s_ in sets])
                row[f"{who}_tflops_warm"], row[f"{who}_tflops_cold"] = round(fl / tw / 1e12, 1), round(fl / tc / 1e12, 1)
            out["shapes"].append(row)
            del sets
    except Exception as e:  # the calibration must never break the benchmark line
        out["error"] = f"{type(e).__name__}: {e}"
        return out
    r0 = out["shapes"][0]
    out.update({"shape": r0["shape"], "hipblaslt_tflops": r0["hipblaslt_tflops_warm"],
                "hipblaslt_frac_of_peak": round(r0["hipblaslt_tflops_warm"] / 2500.0, 4), "mi355x_gemm_tflops": r0["mi355x_gemm_tflops_warm"],
                "note": "top-level numbers: WARM operands, FFN1 shape (round 4's definition); `shapes`: warm and cold, three shapes"})
    return out


def hbm_roofline(model, dev):
    """the HBM-bound half of the step, measured live with HIP events on the kernels' own stream: algorithmic bytes
    (compulsory reads + writes) / time for the three dominant HBM-bound kernels of the step at the
    benchmarked shapes -- LayerNorm forward (fp32 in, bf16 out), fused LayerNorm backward (+ cast of the next operand) and the
    fused AdamW step over the encoder's flat buffer.  (Launch counts per step: 5 LayerNorms x 18 layers, minus the 17 layer
    boundaries where norm_out and the next layer's norm_feed_forward1 run as one kernel, forward and backward.)"""
    from nemo_amd import ops
    enc = model.encoder
    d = enc.d_model
    M = 16032
    x = torch.randn(M, d, device=dev)
    y = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
    mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    ln = enc.layers[0].norm_feed_forward1
    dy = torch.randn(M, d, device=dev).to(torch.bfloat16)
    dres = torch.zeros(M, d, device=dev)
    cast = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
    fp = enc.flat_parameters()
    n = fp.flat.numel()
    m1, m2 = torch.zeros_like(fp.flat), torch.zeros_like(fp.flat)
    wcopy, gcopy = fp.flat.detach().clone(), torch.randn_like(fp.flat) * 1e-3
    cases = {
        "ln_fwd": (lambda: ops.layernorm_fwd(x, ln.weight, ln.bias, y, mean, rstd, M, d, 1e-5), M * d * (4 + 2) + 8 * M, 73),
        "ln_bwd_fused_cast": (lambda: ops.layernorm_bwd(dy, x, ln.weight, mean, rstd, dres, True, ln.weight.grad, ln.bias.grad, M, d,
                                                        cast_out=cast, cast_scale=0.5), M * d * (2 + 4 + 4 + 4 + 2) + 8 * M, 56),
        "adamw": (lambda: ops.adamw_step(wcopy, gcopy, m1, m2, 1e-4, 0.9, 0.98, 1e-8, 1e-3, 1), n * 28, 1),
    }
    out, tot_b, tot_t = {}, 0.0, 0.0
    for name, (fn, nbytes, per_step) in cases.items():
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / reps
        out[name] = {"GBps": round(nbytes / t / 1e9, 1), "us": round(t * 1e6, 1), "bytes": int(nbytes), "launches_per_step": per_step}
        tot_b += nbytes * per_step
        tot_t += t * per_step
    ach = tot_b / tot_t / 1e9
    return {"bound": "hbm", "kernel": "ln_fwd + ln_bwd_fused + adamw (time-weighted over one step)", "achieved": round(ach, 1),
            "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": None, "per_kernel": out}


def self_launch(a):
    """`python bench.py --gpus N` with no launcher around it (the reference gets its ranks from Lightning's own launcher:
    `devices: -1, strategy: ddp`, examples/asr/conf/conformer/conformer_ctc_bpe.yaml:195-201): re-executes this script as N ranks through
    torch.distributed.run on one node, rank -> its own GPU, RCCL.  Fewer visible GPUs than ranks is an error, never a silent N = 1
    line (BENCH_DEVICE=<i> rehearses the control flow with all ranks on one GPU, BENCH_DIST_BACKEND=gloo then carries the collectives)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if "BENCH_DEVICE" not in os.environ and have < a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but only {have} GPU(s) are visible -- refusing to report a smaller job under this flag")
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL / tensor sharing across processes on this driver)
    env["BENCH_SELF_LAUNCHED"] = "1"
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(self_launch(a))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    # BENCH_DEVICE / BENCH_DIST_BACKEND exist to rehearse the N > 1 control flow on a one-GPU box (all ranks on device 0,
    # gloo moving the CUDA tensors through the host); the driver's runs use neither: rank -> its own GPU, RCCL
    rehearsal = "BENCH_DEVICE" in os.environ
    if not rehearsal and torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} (local {local_rank}) has no GPU of its own: {torch.cuda.device_count()} visible")
    local_rank = int(os.environ.get("BENCH_DEVICE", local_rank))
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks_seen = None
    uuid_collision = False
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)
        if torch.distributed.get_world_size() != a.gpus:
            raise SystemExit(f"bench.py: process group has {torch.distributed.get_world_size()} ranks, --gpus {a.gpus}")
        # which physical device every rank sits on (an all-gather of the device identities: N ranks on N distinct GPUs)
        props = torch.cuda.get_device_properties(dev)
        ident = str(getattr(props, "uuid", None) or f"{props.name}#{local_rank}")
        seen = [None] * world
        torch.distributed.all_gather_object(seen, {"rank": rank, "device": local_rank, "uuid": ident})
        ranks_seen = seen
        # hard check: every rank of this node selected a device index of its own.  The UUIDs are reported next to it; a runtime
        # that hands out the same UUID string for distinct devices must not stop the job, so they only decide when they differ
        # in the OTHER direction (distinct indices are given, identical UUIDs are a note in `distributed.ranks_seen`).
        if not rehearsal and len({x["device"] for x in seen}) != world:
            raise SystemExit(f"bench.py: {world} ranks share {len({x['device'] for x in seen})} device index(es): {seen}")
        # identical UUIDs on distinct device indices: either a runtime quirk or ranks sharing a physical GPU.  The run goes on (a quirk
        # must not cost the node's one scaling run), but the line says so: `distributed.valid` is false and the reader decides.
        # BENCH_STRICT_UUID=1 turns the note into a hard stop.
        uuid_collision = (not rehearsal) and len({x["uuid"] for x in seen}) != world
        if uuid_collision:
            if os.environ.get("BENCH_STRICT_UUID", "0") == "1":
                raise SystemExit(f"bench.py: {world} ranks report {len({x['uuid'] for x in seen})} distinct device UUID(s): {seen}")
            if rank == 0:
                print(f"[bench] note: {world} ranks on {world} device indices report {len({x['uuid'] for x in seen})} distinct UUID(s) "
                      f"(line marked distributed.valid=false): {seen}", file=sys.stderr, flush=True)

    from nemo_amd import ops
    from nemo_amd.models import EncDecCTCModel, conformer_ctc_config

    cdt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    if a.model == "transducer":
        from nemo_amd.models import EncDecRNNTModel, fastconformer_transducer_config
        vocab = 1024
        cfg = fastconformer_transducer_config(a.size, vocab_size=vocab, spec_augment=not a.no_spec_augment, compute_dtype=cdt)
        model = EncDecRNNTModel(cfg)
        model.decoder.compute_dtype = model.joint.compute_dtype = cdt
        a.no_cpu_baseline = True  # (the CPU leg times the Conformer-CTC oracle)
    elif a.model == "squeezeformer":
        from nemo_amd.models import squeezeformer_ctc_config
        vocab = 128
        cfg = squeezeformer_ctc_config(a.size, vocab_size=vocab, spec_augment=not a.no_spec_augment, compute_dtype=cdt)
        model = EncDecCTCModel(cfg)
        model.decoder.compute_dtype = cdt
        a.no_cpu_baseline = True  # (the CPU leg times the Conformer-CTC oracle)
    else:
        vocab = 128
        try:
            cfg = conformer_ctc_config(a.size, vocab_size=vocab, spec_augment=not a.no_spec_augment, compute_dtype=cdt)
        except KeyError:
            raise SystemExit(f"bench.py: --size {a.size} is not a Conformer-CTC recipe size (small, medium, large; the others belong "
                             "to --model squeezeformer)") from None
        model = EncDecCTCModel(cfg)
        model.decoder.compute_dtype = cdt
    model = model.to(dev).train()
    model.setup_optimization()
    if a.packed is not None:
        model.encoder.packed_rows = {"auto": "auto", "0": False, "1": True}[a.packed]
    if a.launch is not None:
        model.encoder.use_graphs, model.encoder.graph_auto = a.launch != "live", a.launch == "auto"
    if a.pad_to is not None:
        model.preprocessor.featurizer.pad_to = a.pad_to
    var_info = None
    if a.var_len:
        a.no_roofline = a.no_cpu_baseline = True  # (both describe the fixed-length headline workload)
        batches, valid_secs, var_info = var_len_batches(a, rank, world, vocab, dev)
    else:
        audio, alen, tok, tl = synthetic_batch(a.batch, a.secs, vocab=vocab, seed=1234 + rank)
        batches = [[audio.to(dev), alen.to(dev), tok.to(dev), tl.to(dev)]] * (a.warmup + a.steps)
        valid_secs = [a.batch * a.secs] * (a.warmup + a.steps)
    batch = batches[-1]

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if world > 1:
        # pre-flight: one all-reduce per communicator this run will use, checked for its value, BEFORE any model step -- a broken
        # RCCL / xGMI setup fails here, loudly and by name, instead of as a hang inside the first backward
        for gname, grp in [("world", None)] + [("syncbn", g) for g in {id(m.setup_process_groups()): m.setup_process_groups()
                                                                     for m in model.trainable_modules()
                                                                     if hasattr(m, "setup_process_groups")}.values()
                                                if g is not None and g is not torch.distributed.group.WORLD]:
            for dt_ in (torch.float32, torch.float64):
                t = torch.full((1024,), float(rank + 1), device=dev, dtype=dt_)
                torch.distributed.all_reduce(t, group=grp)
                torch.cuda.synchronize()
                want = world * (world + 1) / 2.0
                if not bool((t == want).all()):
                    raise SystemExit(f"bench.py: pre-flight all-reduce on the '{gname}' communicator returned {t[0].item()}, expected {want}")
        # ... and one exchange per statistics mailbox (MI355X_SYNCBN_MAILBOX=1: peer-mapped memory instead of a communicator)
        for m in model.trainable_modules():
            mbx = getattr(m, "_syncbn_mailbox", None)
            if mbx is not None:
                t = torch.full((1025,), float(rank + 1), device=dev, dtype=torch.float64)
                mbx.all_reduce_(t)
                torch.cuda.synchronize()
                done, missing, kind = mbx.status()
                if missing or not bool((t == world * (world + 1) / 2.0).all()):
                    raise SystemExit(f"bench.py: pre-flight exchange over the statistics mailbox failed on rank {rank} "
                                     f"(value {t[0].item()}, missing rank {missing - 1 if missing else None}, memory kind {kind})")
        if rank == 0:
            print(f"[bench] pre-flight all-reduce ok on {world} ranks ({backend})", file=sys.stderr, flush=True)
    loss = None
    for i in range(a.warmup):
        loss = model.fit_step(batches[i])["loss"]
    # the encoder times live launches against hipGraph replay per shape and keeps the faster (MI355X_GRAPHS=auto): more un-timed
    # steps until that trial is over, so that no recording / trial step falls into the timed region (fixed-shape runs only: a
    # variable-length run never repeats a shape often enough to record it)
    warmup_extra = 0
    settled = getattr(model.encoder, "graphs_settled", lambda: True)
    while not a.var_len and warmup_extra < 16:
        torch.cuda.synchronize()
        done = settled() and warmup_extra >= 1
        if world > 1:  # every rank runs the same number of steps (a step holds collectives): stop only when ALL ranks are settled
            flag = torch.tensor([0.0 if done else 1.0], device=dev)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
            done = flag.item() == 0.0
        if done:
            break
        loss = model.fit_step(batches[-1])["loss"]
        warmup_extra += 1
    previsit = None
    if a.var_len and getattr(model.encoder, "use_graphs", False) and world == 1:
        # Variable-length run with recorded launch sequences: visit every padded length of the timed batches (un-timed) until its
        # sequence is recorded / decided -- the state a training run is in after its first pass over the duration buckets.  The
        # timed steps then run what a long run runs; lengths that could not be recorded stay on live launches and are counted.
        enc = model.encoder
        hop = int(round(model.preprocessor.featurizer.hop_length)) if hasattr(model.preprocessor.featurizer, "hop_length") else 160
        pt = int(getattr(model.preprocessor.featurizer, "pad_to", 0) or 0)
        def padded_frames(bt):
            t = bt[0].shape[1] // hop + 1
            return -(-t // pt) * pt if pt > 0 else t
        by_len = {}
        for bt in batches[a.warmup:]:
            by_len.setdefault(padded_frames(bt), bt)
        previsit = {"padded_lengths": len(by_len), "steps": 0}
        if len(by_len) <= getattr(enc, "max_graph_sets", 8):
            enc._graph_sets.clear()   # (the warm-up batches' lengths: seen once or twice, never again)
            if enc.graph_auto:        # the longest length runs its live-vs-recorded trial first; the others take its decision over
                for _ in range(16):
                    model.fit_step(by_len[max(by_len)])
                    previsit["steps"] += 1
                    torch.cuda.synchronize()
                    if enc.graphs_settled():
                        break
            for rnd in range(5):
                for tq, bt in sorted(by_len.items(), reverse=True):
                    model.fit_step(bt)
                    previsit["steps"] += 1
                torch.cuda.synchronize()
                if enc.graphs_recorded() and rnd >= 1:
                    break
            previsit["all_recorded"] = bool(enc.graphs_recorded())
        enc.replayed_steps = enc.live_steps = 0
    barrier()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)] if a.var_len else None
    t0 = time.perf_counter()
    for i in range(a.steps):
        if marks:
            marks[i].record()
        loss = model.fit_step(batches[a.warmup + i])["loss"]
    if marks:
        marks[-1].record()
    host_s = time.perf_counter() - t0  # host time to ISSUE the K steps (the GPU may still be running the last ones)
    barrier()
    dt = time.perf_counter() - t0
    timed_valid = sum(valid_secs[a.warmup:])
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    ms = dt / a.steps * 1e3
    if a.var_len:
        tv = torch.tensor([timed_valid], device=dev, dtype=torch.float64)
        if world > 1:
            torch.distributed.all_reduce(tv)
        value = tv.item() / dt  # VALID audio seconds of all ranks per wall-clock second
        per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
        pl = getattr(model.encoder, "packed_last", None)
        var_info.update({"packed_rows": str(getattr(model.encoder, "packed_rows", False)),
                         "last_step_rows_packed_vs_padded": list(pl) if pl else None,
                         "valid_audio_seconds_timed": round(tv.item(), 1),
                         "ms_per_step_min_median_max": [round(per_step[0], 2), round(per_step[len(per_step) // 2], 2), round(per_step[-1], 2)]})
    else:
        value = world * a.batch * a.secs / (dt / a.steps)
    final_loss = float(loss.item())

    # ---- what the gradient exchange looked like (diagnosable SCALE runs): collective library, buckets, exposed time
    dist_info = {"world_size": world, "backend": backend if world > 1 else None, "ranks_seen": ranks_seen,
                 # valid = a scaling number may be read off this line: false when ranks on distinct device indices reported the same
                 # device UUID, and false for a rehearsal (BENCH_DEVICE: every rank on ONE GPU -- control flow only)
                 "valid": not uuid_collision and not (rehearsal and world > 1), "rehearsal": bool(rehearsal and world > 1),
                 "launcher": "self (torch.distributed.run)" if os.environ.get("BENCH_SELF_LAUNCHED") else
                 ("external" if world > 1 else None)}
    try:
        v = torch.cuda.nccl.version()
        dist_info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception as e:  # noqa: BLE001
        dist_info["rccl_version"] = f"unavailable ({type(e).__name__})"
    if world > 1:
        syncs = model._grad_syncs()
        for gs in syncs:
            gs.profile = True
        enc = model.encoder
        enc.syncbn_profile = []
        use_g, enc.use_graphs = getattr(enc, "use_graphs", False), False  # (the instrumented steps run on live launches)
        for _ in range(2):
            enc.syncbn_profile.clear()
            model.fit_step(batch)
        barrier()
        enc.use_graphs = use_g
        exposed = sum(gs.exposed_events[0].elapsed_time(gs.exposed_events[1]) for gs in syncs if gs.exposed_events)
        syncbn_ms = sum(e0.elapsed_time(e1) for e0, e1 in enc.syncbn_profile)
        n_syncbn = len(enc.syncbn_profile)
        enc.syncbn_profile = None
        for gs in syncs:
            gs.profile = False
        t = torch.tensor([exposed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dist_info.update({
            "grad_allreduce_launches_per_step": sum(gs.launches_last_step for gs in syncs),
            "grad_bytes_per_step": int(sum(gs.grad.numel() * 4 for gs in syncs)),
            "bucket_bytes": int(syncs[0].bucket_elems * 4),
            "exposed_exchange_ms_max_over_ranks": round(float(t.item()), 3),
            "syncbn_allreduces_per_step": n_syncbn,
            "syncbn_exposed_ms_this_rank": round(syncbn_ms, 3),
            "grad_wire_dtype": "bf16" if syncs[0].wire_dtype is not None else "fp32",
            # every all-reduce of the last instrumented step on this rank, in issue order: [MiB, ms on the exchange stream from its
            # enqueue to its completion]; the last entry of the encoder's list is the tail bucket (cut to <= tail_bytes)
            "buckets_mib_ms_this_rank": [[round(n_ * 4 / 2 ** 20, 2), round(e0.elapsed_time(e1), 3)]
                                         for gs in syncs for (n_, e0, e1) in gs.bucket_events_last_step],
            "tail_bytes": int(syncs[0].tail_elems * 4),
            "syncbn_own_process_group": os.environ.get("MI355X_SYNCBN_OWN_GROUP", "0") == "1",
            # how the statistics travel: "mailbox" = one kernel launch per exchange over peer-mapped memory (csrc/mailbox.hip),
            # "process group" = torch.distributed.all_reduce
            "syncbn_exchange": ("mailbox" if getattr(model.encoder, "_syncbn_mailbox", None) is not None else "process group"),
        })

    roof = None
    if not a.no_roofline:
        # per-launch durations need the kernels one after another: the weight-gradient side stream (which overlaps wgrad
        # GEMMs with the HBM-bound kernels in the timed steps above) is switched off for this one instrumented step.
        # EVERY rank runs the step (it contains the SyncBN and gradient collectives); only rank 0 records and reports.
        side = getattr(model.encoder, "wgrad_side_stream", False)
        in_step_records = None
        if side:
            # the same events with the side stream ON, i.e. the configuration the timed steps ran in: a launch's duration then
            # includes whatever shared the chip with it (`frac_in_step`, next to the serialised `frac`)
            if rank == 0:
                ops.GEMM_PROFILE = []
            model.fit_step(batch)
            barrier()
            if rank == 0:
                in_step_records, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
        model.encoder.wgrad_side_stream = False
        if rank == 0:
            ops.GEMM_PROFILE = []
        model.fit_step(batch)
        barrier()
        model.encoder.wgrad_side_stream = side
    if rank == 0 and not a.no_roofline:
        def aggregate(records):
            out_ = {}
            for (variant, M, N, K, nb, e0, e1) in records:
                g = out_.setdefault(variant, [0.0, 0.0, 0])
                g[0] += 2.0 * M * N * K * nb
                g[1] += e0.elapsed_time(e1) * 1e-3
                g[2] += 1
            return out_
        agg = aggregate(ops.GEMM_PROFILE)
        agg_in_step = aggregate(in_step_records) if in_step_records else None
        table_path = os.environ.get("BENCH_GEMM_TABLE")
        if table_path:
            shapes = {}
            for (variant, M, N, K, nb, e0, e1) in ops.GEMM_PROFILE:
                g = shapes.setdefault((variant, M, N, K, nb), [0.0, 0])
                g[0] += e0.elapsed_time(e1) * 1e-3
                g[1] += 1
            with open(table_path, "w") as f:
                f.write("variant M N K batch launches total_ms avg_us TFLOP/s\n")
                for (variant, M, N, K, nb), (t, c) in sorted(shapes.items(), key=lambda kv: -kv[1][0]):
                    f.write(f"{variant} {M} {N} {K} {nb} {c} {t*1e3:.3f} {t/c*1e6:.1f} {2.0*M*N*K*nb*c/t/1e12:.1f}\n")
        ops.GEMM_PROFILE = None
        dom = max(agg.items(), key=lambda kv: kv[1][1])
        flops, secs_, cnt = dom[1]
        peak = 2500.0 if a.dtype == "bf16" else 157.3
        # HBM bytes per launch of that kernel family from the TCC counters: collected off-line (PMC passes serialise the
        # kernels, so they cannot run inside the timed benchmark) and committed with its method under profiles/
        traffic = mfma_busy = None
        traffic_note = "not reported: no PMC table for this build of the GEMM sources"
        try:
            import glob
            tfile = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_traffic.json")))[-1]  # the latest round's table
            with open(tfile) as f:
                tj = json.load(f)
            if tj.get("source_sha256_16") != _source_hash():
                traffic_note = (f"not reported: profiles/{os.path.basename(tfile)} was measured on other GEMM sources "
                                "(hash mismatch)")
            elif tj.get("kernel") == f"gemm_{dom[0]}" and a.size == "large" and a.batch == 32 and a.dtype == "bf16":
                traffic = tj["traffic_bytes_per_launch"]
                mfma_busy = tj.get("mfma_busy_frac")
                traffic_note = tj.get("method")
        except (OSError, ValueError, KeyError):
            traffic = None
        roof = {"bound": "mfma", "kernel": f"gemm_{dom[0]}", "achieved": round(flops / secs_ / 1e12, 1), "peak": peak,
                "unit": "TFLOP/s", "frac": round(flops / secs_ / 1e12 / peak, 4), "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, off-line; stamped with the hash "
                                "of the GEMM sources it was measured on)",
                "traffic_note": traffic_note,
                "mfma_busy_frac_pmc": mfma_busy,  # SQ_VALU_MFMA_BUSY_CYCLES / (1024 x GRBM_GUI_ACTIVE), same table
                "launches_per_step": cnt, "avg_launch_us": round(secs_ / cnt * 1e6, 1),
                # the same kernel family timed inside the step as the timed region runs it (weight-gradient side stream ON: the
                # launches share CUs with the side stream's GEMMs); None when the encoder has no side stream
                "frac_in_step": (round(agg_in_step[dom[0]][0] / agg_in_step[dom[0]][1] / 1e12 / peak, 4)
                                 if agg_in_step and dom[0] in agg_in_step else None),
                "achieved_in_step": (round(agg_in_step[dom[0]][0] / agg_in_step[dom[0]][1] / 1e12, 1)
                                     if agg_in_step and dom[0] in agg_in_step else None),
                # SURVEY 8(d)'s formula charges linear_pos (P x d x d) per UTTERANCE: 18.24 TFLOP per step; the code computes it once
                # per BATCH (0.03 instead of 0.91 TFLOP): 17.36 TFLOP of work actually done -- both are printed
                "whole_step_frac": (round(18.24e12 / (ms * 1e-3) / 1e12 / peak, 4)
                                    if a.model == "ctc" and a.size == "large" and a.batch == 32 and a.secs == 20.0 and world == 1 else None),
                "whole_step_frac_linear_pos_once_per_batch": (round(17.36e12 / (ms * 1e-3) / 1e12 / peak, 4)
                                    if a.model == "ctc" and a.size == "large" and a.batch == 32 and a.secs == 20.0 and world == 1 else None),
                "gemm_time_share_of_step": round(sum(v[1] for v in agg.values()) * 1e3 / ms, 3),
                "all_variants": {k: {"tflops": round(v[0] / v[1] / 1e12, 1), "ms": round(v[1] * 1e3, 2), "launches": v[2]}
                                 for k, v in agg.items()}}

        if a.dtype == "bf16":
            roof["vendor_library_same_box"] = vendor_ceiling(dev)

    roof_hbm = None
    if rank == 0 and not a.no_roofline and a.size == "large" and a.model == "ctc":
        roof_hbm = hbm_roofline(model, dev)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.size, a.secs, vocab, a.cpu_batch, a.cpu_steps)

    if rank == 0:
        line = {
            "metric": ("audio-sec/s training throughput, Conformer-CTC-Large @ 1/2/4/8 MI355X" if a.size == "large"
                       else f"audio-sec/s training throughput, Conformer-CTC-{a.size}") if a.model == "ctc"
            else f"audio-sec/s training throughput, FastConformer-Transducer-{a.size.capitalize()}" if a.model == "transducer"
            else f"audio-sec/s training throughput, Squeezeformer-CTC-{a.size.capitalize()}",
            "value": round(value, 1), "unit": "audio-sec/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": (f"Conformer-CTC-{a.size.capitalize()} {a.dtype}, batch={a.batch}x{a.secs:g}s synthetic 16 kHz "
                                    f"clips per GPU, {world}xMI355X (BASELINE.json configs[{1 if world == 1 else 2}])")
                       if a.model == "ctc" else
                       (f"Squeezeformer-CTC-{a.size.capitalize()} {a.dtype} (dw_striding x4, time reduction / recovery), "
                        f"batch={a.batch}x{a.secs:g}s per GPU, {world}xMI355X (BASELINE.json configs[4])") if a.model == "squeezeformer" else
                       (f"FastConformer-Transducer-{a.size.capitalize()} {a.dtype} (x8 dw_striding, LSTM prediction net, fused joint "
                        f"+ RNN-T loss, vocab 1024), batch={a.batch}x{a.secs:g}s per GPU, {world}xMI355X (BASELINE.json configs[3])"),
                       "global_batch": world * a.batch, "clip_seconds": a.secs, "parallelism": f"dp{world}",
                       "step": ("fwd+joint+RNNT-loss" if a.model == "transducer" else "fwd+CTC") + "+bwd+grad-allreduce+AdamW, train mode (dropout, dither, SyncBN"
                               + (")" if a.no_spec_augment else ", SpecAugment)"),
                       "final_loss": round(final_loss, 4)},
        }
        if var_info is not None:
            line["config"]["workload"] += f"; VARIABLE LENGTH: {var_info['durations']}, batches by {var_info['sampler']}"
            line["config"]["variable_length"] = var_info
        if roof is not None:
            line["roofline"] = roof
        if roof_hbm is not None:
            line["roofline_hbm"] = roof_hbm
        ginfo = model.encoder.graph_info() if hasattr(model.encoder, "graph_info") else []
        replayed = any(g.get("fwd_graphs") and str(g.get("decided")) != "eager" for g in ginfo)
        taped = replayed and any(g.get("replay") == "launch tape" for g in ginfo)
        line["launch"] = {"mode": (("encoder forward/backward re-issued from launch tapes (csrc/tape.hip: the recorded nodes as live "
                                    "launches from one C loop)" if taped else
                                    "encoder forward/backward replayed from hipGraph segments (nemo_amd/graphs.py)") +
                                   "; front end, decoder, loss and optimizer launched live") if replayed else
                                  ("every kernel launched live from the Python sequencer" +
                                   (" (the recorded sequence was timed on this box and was slower)" if ginfo else "")),
                          "host_issue_ms_per_step": round(host_s / a.steps * 1e3, 2), "recorded": ginfo,
                          "untimed_steps_beyond_warmup": warmup_extra}
        if previsit is not None:
            previsit.update({"timed_steps_replayed": getattr(model.encoder, "replayed_steps", None),
                             "timed_steps_live": getattr(model.encoder, "live_steps", None)})
            line["launch"]["variable_length_previsit"] = previsit
            line["launch"]["recorded"] = [{k: g.get(k) for k in ("mel_shape", "decided", "auto", "replay")} for g in ginfo]
        line["distributed"] = dist_info
        if cpu is not None:
            line["cpu_baseline"] = cpu
            line["gpu_over_cpu"] = round(value / cpu["value"], 1)
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
