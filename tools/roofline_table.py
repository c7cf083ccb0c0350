"""One table per round from the two PMC summaries (tools/pmc_assemble.py): every kernel of the step with its serialised time per
step, its HBM rate against 8 TB/s and its MFMA utilisation -- the kernels furthest below BOTH roofs first.
    python tools/roofline_table.py profiles r4 3      (3 = optimizer steps inside the PMC runs: `bench.py --steps 1 --warmup 1` + set-up step)"""
import re
import sys


def rows(path, ncol):
    out = {}
    for line in open(path):
        p = [c.strip() for c in line.strip().strip("|").split("|")]
        if len(p) == ncol and p[0].startswith("`") and p[1].isdigit():
            out[p[0].strip("`")] = p[1:]
    return out


def short(k):
    m = re.match(r"_Z\d+([A-Za-z0-9_]+?)(?:I|Pv|P[KF]|v$|\.kd)", k)
    return (m.group(1) if m else k)[:40] + ("<" + "".join(re.findall(r"L[bi](\d)E", k)) + ">" if "ILb" in k or "ILi" in k else "")


def main(d, tag, steps):
    steps = int(steps)
    hbm = rows(f"{d}/{tag}_pmc_hbm_traffic.md", 6)
    mf = rows(f"{d}/{tag}_pmc_mfma_utilisation.md", 6)
    tab = []
    for k, (n, rd, wr, us, gbs) in hbm.items():
        if "copyBuffer" in k:
            continue  # (one-off parameter flattening of the first step, not per-step work: see the per-step kernel stats)
        n, rd, wr, us, gbs = int(n), float(rd), float(wr), float(us), float(gbs)
        busy = float(mf[k][3].rstrip(" %")) if k in mf else 0.0
        tab.append((us * n / steps, k, n / steps, us, rd + wr, gbs / 8000.0, busy / 100.0))
    tab.sort(reverse=True)
    tot = sum(t[0] for t in tab)
    with open(f"{d}/{tag}_roofline_per_kernel.md", "w") as o:
        o.write(f"# Every kernel of the step against both roofs ({tag}; from `{tag}_pmc_hbm_traffic.md` and `{tag}_pmc_mfma_utilisation.md`)\n\n"
                "Serialised times (PMC passes run the kernels one after another, weight-gradient side stream off; Conformer-CTC-Large bf16,\n"
                "B = 32 x 20 s).  HBM fraction = (FETCH_SIZE x 2 + WRITE_SIZE) / time / 8 TB/s; MFMA = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x\n"
                "GRBM_GUI_ACTIVE).  `best` = the larger of the two: how close the kernel is to the roof that could bound it.  Kernels under 5 us\n"
                "per step are summed in the last row.\n\n"
                "| kernel | launches / step | us / launch | us / step | share | MB / launch | HBM frac | MFMA busy | best |\n"
                "|---|---:|---:|---:|---:|---:|---:|---:|---:|\n")
        rest = 0.0
        for t, k, n, us, mb, hf, mu in tab:
            if t < 5.0:
                rest += t
                continue
            o.write(f"| `{short(k)}` | {n:.0f} | {us:.1f} | {t:.0f} | {100 * t / tot:.1f} % | {mb:.1f} | {hf:.2f} | {mu:.2f} | {max(hf, mu):.2f} |\n")
        o.write(f"| (kernels under 5 us per step) | | | {rest:.0f} | {100 * rest / tot:.1f} % | | | | |\n")
        o.write(f"| **total, serialised** | | | **{tot:.0f}** | | | | | |\n")
        w = sum(t[0] * max(t[5], t[6]) for t in tab) / tot
        o.write(f"\nTime-weighted `best` over the step: **{w:.2f}** -- the step as a whole sits at that fraction of whichever roof bounds each kernel.\n")
    print(open(f"{d}/{tag}_roofline_per_kernel.md").read()[:3500])


if __name__ == "__main__":
    main(*sys.argv[1:4])
