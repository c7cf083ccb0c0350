"""Assemble the round's PMC evidence from the three `tools/pmc_dump.py` files of `tools/run_profiles_r3.sh`:
  profiles/r3_pmc_hbm_traffic.md      per-kernel HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE) with the calibration rows
  profiles/r3_pmc_mfma_utilisation.md SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE) per kernel
  profiles/r3_gemm_traffic.json       what bench.py reads for `roofline.traffic` (stamped with the GEMM source hash)
usage: python tools/pmc_assemble.py gpurun_out/r2prof profiles r2"""
import json
import os
import sys

KNOWN = {  # compulsory bytes of kernels whose traffic is known exactly (Large, B = 32 x 20 s: M = 16032 rows, d = 512)
    "ln_fwd_reg_kernelIft": ("LayerNorm fwd f32 -> bf16", 16032 * 512 * 4, 16032 * 512 * 2 + 2 * 16032 * 4),
    "adamw_kernel": ("fused AdamW (p, g, m, v read; p, m, v written; 121.5 M params incl. alignment)", None, None),
}


def main(src, dst, tag):
    F = json.load(open(os.path.join(src, "pmc_fetch.pmc.json")))
    W = json.load(open(os.path.join(src, "pmc_write.pmc.json")))
    Mf = json.load(open(os.path.join(src, "pmc_mfma.pmc.json")))
    src_hash = open(os.path.join(src, "source_hash.txt")).read().strip()
    rows = []
    for k, v in F.items():
        f = v.get("FETCH_SIZE")
        w = W.get(k, {}).get("WRITE_SIZE")
        if not f or not w:
            continue
        n = f["dispatches"]
        rd = 2.0 * f["sum"] / n * 1024  # gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section)
        wr = w["sum"] / w["dispatches"] * 1024
        rows.append((k, n, rd, wr, f["mean_us"]))
    rows.sort(key=lambda r: -(r[2] + r[3]) * r[1])
    with open(os.path.join(dst, f"{tag}_pmc_hbm_traffic.md"), "w") as o:
        o.write(f"# HBM traffic per kernel launch from the TCC counters (rocprofv3 PMC), GEMM sources {src_hash}\n\n"
                "Two separate passes, one counter each (`tools/run_profiles_r3.sh`; `--kernel-trace --pmc X` only, weight-gradient\n"
                "side stream off so that every row is one kernel on its own; `bench.py --steps 1 --warmup 1`, i.e. two optimizer\n"
                "steps of Conformer-CTC-Large bf16, B = 32 x 20 s).  Read bytes = FETCH_SIZE (KiB) x 1024 x 2 (gfx950 tallies 128-B\n"
                "read requests at 64 B), written bytes = WRITE_SIZE (KiB) x 1024; one counter sample per dispatch in this rocprofv3.\n\n"
                "Calibration on kernels with known compulsory traffic (same run):\n\n")
        for key, (what, rb, wb) in KNOWN.items():
            for k, n, rd, wr, us in rows:
                if key in k and rb:
                    o.write(f"* `{key}` ({what}): compulsory read {rb/1e6:.1f} MB / written {wb/1e6:.1f} MB; counters {rd/1e6:.1f} / "
                            f"{wr/1e6:.1f} MB\n")
        o.write("\n| kernel | launches | HBM read MB / launch | HBM written MB / launch | mean us (under PMC) | GB/s |\n|---|---:|---:|---:|---:|---:|\n")
        for k, n, rd, wr, us in rows:
            o.write(f"| `{k[:72]}` | {n} | {rd/1e6:.1f} | {wr/1e6:.1f} | {us:.1f} | {(rd+wr)/us/1e3:.0f} |\n")
    with open(os.path.join(dst, f"{tag}_pmc_mfma_utilisation.md"), "w") as o:
        o.write(f"# MFMA utilisation per kernel (rocprofv3 PMC), GEMM sources {src_hash}\n\n"
                "`rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` (one pass, SQ and GRBM blocks are independent),\n"
                "same command as the traffic table.  busy = sum of SQ_VALU_MFMA_BUSY_CYCLES over the dispatch's samples (= 32 cycles x\n"
                "number of 32x32x16 MFMAs: checked on the 16032 x 2048 x 512 launch = 33.0 M); GUI cycles = GRBM_GUI_ACTIVE summed over its\n"
                "8 samples (one per XCD) / 8; utilisation = busy / (1024 SIMDs x GUI cycles).\n\n"
                "| kernel | launches | MFMA busy Mcycles / launch | GUI kcycles / launch | MFMA utilisation | mean us |\n|---|---:|---:|---:|---:|---:|\n")
        mrows = []
        for k, v in Mf.items():
            b, g = v.get("SQ_VALU_MFMA_BUSY_CYCLES"), v.get("GRBM_GUI_ACTIVE")
            if not b or not g or b["sum"] == 0:
                continue
            n = b["dispatches"]
            gui = g["sum"] / n / (g["samples"] / n)
            mrows.append((k, n, b["sum"] / n, gui, b["sum"] / n / (1024.0 * gui), b["mean_us"]))
        mrows.sort(key=lambda r: -r[2] * r[1])
        for k, n, busy, gui, frac, us in mrows:
            o.write(f"| `{k[:72]}` | {n} | {busy/1e6:.2f} | {gui/1e3:.1f} | {100*frac:.1f} % | {us:.1f} |\n")
        lm = [v for k, v in Mf.items() if "logmel" in k]
        if lm:
            o.write("\n`logmel_kernel` issues no MFMA (busy 0): it is a VALU radix FFT, see the traffic table for its HBM side.\n")
    # ---- the NT family (what bench.py's GEMM_PROFILE calls bf16_NT: !transA && !transB)
    def is_nt(name):  # v2 / v4 <TA = 0, TB = 0, ..>, the persistent v5, v8 <G, TN = 0, AH>
        return ("gemm_bf16_v5" in name or "ILb0ELb0E" in name or ("gemm_bf16_v8" in name and "ELb0ELi" in name)) and "grouped" not in name
    nt = [r for r in rows if is_nt(r[0])]
    n_l = sum(r[1] for r in nt)
    traffic = sum((r[2] + r[3]) * r[1] for r in nt) / n_l
    mnt = [r for r in mrows if is_nt(r[0])]
    busy = sum(r[2] * r[1] for r in mnt) / sum(1024.0 * r[3] * r[1] for r in mnt)
    json.dump({"kernel": "gemm_bf16_NT", "traffic_bytes_per_launch": int(traffic), "launches_sampled": n_l,
               "source_sha256_16": src_hash,
               "method": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950), mean over the {n_l} NT "
                         f"launches of two steps, see profiles/{tag}_pmc_hbm_traffic.md",
               "mfma_busy_frac": round(busy, 4),
               "mfma_method": f"sum(SQ_VALU_MFMA_BUSY_CYCLES) / (1024 SIMDs x GRBM_GUI_ACTIVE) over the same launches, see "
                              f"profiles/{tag}_pmc_mfma_utilisation.md"},
              open(os.path.join(dst, f"{tag}_gemm_traffic.json"), "w"), indent=1)
    print("NT launches", n_l, "traffic MB/launch", traffic / 1e6, "mfma busy", busy)


if __name__ == "__main__":
    main(*sys.argv[1:4])
