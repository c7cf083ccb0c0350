"""Per-step launch counts and kernel time from two `rocprofv3 --kernel-trace --stats` runs of bench.py that differ only in
--steps (tools/run_profiles_r3.sh: 4 and 12 timed steps, 2 warm-up steps each): per step = (run B - run A) / (steps_B - steps_A);
what is left over in run A after subtracting its 6 steps is set-up work (parameter flattening, first-touch allocations).
usage: python tools/per_step_stats.py A.csv stepsA B.csv stepsB out.md"""
import csv
import sys


def load(p):
    return {r["kernel"]: (int(r["calls"]), int(r["total_ns"])) for r in csv.DictReader(open(p))}


def main(pa, sa, pb, sb, out):
    a, b = load(pa), load(pb)
    sa, sb = int(sa), int(sb)
    rows = []
    for k in sorted(set(a) | set(b)):
        ca, ta = a.get(k, (0, 0))
        cb, tb = b.get(k, (0, 0))
        per_c, per_t = (cb - ca) / (sb - sa), (tb - ta) / (sb - sa)
        rows.append((k, per_c, per_t / 1e3, ca - sa * per_c))
    rows.sort(key=lambda r: -r[2])
    tot_t, tot_c = sum(r[2] for r in rows), sum(r[1] for r in rows)
    with open(out, "w") as o:
        o.write(f"# Kernel time per optimizer step (Conformer-CTC-Large bf16, B = 32 x 20 s), from {pa} ({sa} steps) and {pb} ({sb} steps)\n\n"
                "`rocprofv3 --kernel-trace --stats -- python bench.py --steps N --warmup 2 --no-cpu-baseline --no-roofline`; both streams\n"
                "active (the weight-gradient stream overlaps the main chain, so the per-kernel durations include co-running kernels and\n"
                "their sum exceeds the wall-clock step).  `set-up launches` = launches of the shorter run not explained by its steps.\n\n"
                "| kernel | launches / step | us / step | set-up launches |\n|---|---:|---:|---:|\n")
        for k, c, t, s in rows:
            if c < 0.01 and abs(s) < 0.5:
                continue
            o.write(f"| `{k[:84]}` | {c:.1f} | {t:.1f} | {s:.0f} |\n")
        o.write(f"| **total** | {tot_c:.0f} | {tot_t:.0f} | |\n")
    cp = [r for r in rows if "copyBuffer" in r[0]]
    print("per step: launches", tot_c, "us", tot_t, "copyBuffer/step", cp and cp[0][1], "setup", cp and cp[0][3])


if __name__ == "__main__":
    main(*sys.argv[1:6])
