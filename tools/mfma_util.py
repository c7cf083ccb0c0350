"""MFMA utilisation per kernel from a tools/pmc_summary.py table that holds SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE rows.

utilisation = sum(SQ_VALU_MFMA_BUSY_CYCLES over the samples of a dispatch) / (1024 SIMDs x GRBM_GUI_ACTIVE of the dispatch)
(the busy counter advances 32 cycles per v_mfma_f32_32x32x16_bf16 on the SIMD that issues it, MI355X_MICROARCH.md; 32 cycles
is also that instruction's issue interval, so 100 % = every SIMD issuing MFMAs back to back = the 2.5 PFLOP/s dense peak).
GRBM_GUI_ACTIVE is reported once per XCD (8 samples per dispatch, each the dispatch's duration in cycles)."""
import re
import sys

rows = {}
for line in open(sys.argv[1]):
    m = re.match(r"\| `(.+?)` \| (\w+) \| (\d+) \| ([\d.]+) \| ([\d.]+) \|", line)
    if m:
        rows.setdefault(m.group(1), {})[m.group(2)] = (int(m.group(3)), float(m.group(4)), float(m.group(5)))
print("| kernel | dispatches | MFMA busy cycles / dispatch (all SIMDs) | GPU cycles / dispatch | MFMA utilisation | mean µs (under PMC) |")
print("|---|---:|---:|---:|---:|---:|")
out = []
for k, v in rows.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
        nb, mb, us = v["SQ_VALU_MFMA_BUSY_CYCLES"]
        ng, mg, _ = v["GRBM_GUI_ACTIVE"]
        disp = ng / 8.0
        busy = nb * mb / disp
        out.append((busy * disp, k, disp, busy, mg, busy / (1024.0 * mg), us))
for _, k, disp, busy, mg, u, us in sorted(out, reverse=True):
    print(f"| `{k[:70]}` | {disp:.0f} | {busy:,.0f} | {mg:,.0f} | {100*u:.1f} % | {us:.1f} |")
