"""Static instruction mix of the hottest loop of a kernel (no GPU needed): compile one .hip source for gfx950, find every loop of the
named kernel (a backward branch to a label), and print for the loops with MFMAs -- per iteration -- the MFMA count and its matrix-pipe
cycles, VALU / transcendental / packed-convert counts with their issue cycles, LDS and global memory instructions, SALU, barriers
and every s_waitcnt.  One wave issues one instruction per ~4 cycles (wave64 on a 16-lane SIMD; transcendentals 8-16), a 32x32x16 bf16
MFMA occupies the SIMD's matrix pipe for 32 cycles (16 for 16x16x32): comparing the two columns says whether a loop can be MFMA-bound
at all, and by how much the other streams exceed it.
    python tools/loop_mix.py nemo_amd/csrc/attention.hip relpos_flash_fwd_kernel [relpos_flash_bwd_dq_kernel ...]
LOOP_MIX_FLAGS="-fno-slp-vectorize" adds compiler flags (what-if runs), LOOP_MIX_HIST=0 drops the opcode histograms."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def asm_of(src):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-S", "--cuda-device-only", "-I",
                        os.path.join(ROOT, "include"), "-I", os.path.dirname(os.path.abspath(src))] +
                       os.environ.get("LOOP_MIX_FLAGS", "").split() + [src, "-o", out], check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_cvt_pk"):
        return "cvt_pk"
    if op.startswith(("ds_read", "ds_load")):
        return "lds_read"
    if op.startswith(("ds_write", "ds_store")):
        return "lds_write"
    if op.startswith(("ds_bpermute", "ds_permute", "ds_swizzle")):
        return "lds_permute"
    if op.startswith("ds_"):
        return "lds_other"
    if op.startswith(("global_load_lds", "buffer_load") ) and "lds" in op:
        return "lds_dma"
    if op.startswith(("global_load", "buffer_load", "flat_load")):
        return "vmem_load"
    if op.startswith(("global_store", "buffer_store", "flat_store", "global_atomic", "buffer_atomic")):
        return "vmem_store"
    if op == "s_barrier":
        return "barrier"
    if op == "s_waitcnt":
        return "waitcnt"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    return "other"


def loops_of(body):
    lines = body.splitlines()
    label_at = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            label_at[m.group(1)] = i
    loops = []
    for i, ln in enumerate(lines):
        m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln) or re.match(r"^\s+s_branch\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in label_at and label_at[m.group(1)] < i:
            loops.append((label_at[m.group(1)], i))
    # innermost first: drop loops that contain another loop entirely only when asked; keep all, sorted by MFMA count later
    return lines, loops


def histogram(lines, lo, hi, kinds=("valu", "trans", "cvt_pk", "salu")):
    h = {}
    for ln in lines[lo:hi + 1]:
        if not ln.startswith("\t") or ln.strip().startswith((";", ".")):
            continue
        op = ln.split()[0]
        if classify(op) in kinds:
            h[op] = h.get(op, 0) + 1
    return sorted(h.items(), key=lambda kv: -kv[1])


def mix(lines, lo, hi):
    c, waits = {}, []
    for ln in lines[lo:hi + 1]:
        if not ln.startswith("\t") or ln.strip().startswith((";", ".")):
            continue
        parts = ln.split()
        op = parts[0]
        k = classify(op)
        if k == "mfma":
            k = "mfma32" if "32x32" in op else "mfma16"
        c[k] = c.get(k, 0) + 1
        if k == "waitcnt":
            waits.append(" ".join(parts[1:]))
        if k == "lds_dma" or ("lds" in op and op.startswith("global_load")):
            c["lds_dma"] = c.get("lds_dma", 0) + (0 if k == "lds_dma" else 1)
    return c, waits


def main(src, names):
    txt = asm_of(src)
    for k in re.split(r"\n(?=_Z[\w]+:\s)", txt):
        m = re.match(r"(_Z[\w]+):", k)
        if not m or not any(n in m.group(1) for n in names):
            continue
        body = k.split(".Lfunc_end")[0]
        lines, loops = loops_of(body)
        rows = []
        for lo, hi in loops:
            c, waits = mix(lines, lo, hi)
            nm = c.get("mfma32", 0) + c.get("mfma16", 0)
            if nm:
                rows.append((nm, lo, hi, c, waits))
        rows.sort(reverse=True)
        print(f"## `{m.group(1)[:70]}`: {len(loops)} loops, {len(rows)} with MFMAs\n")
        for nm, lo, hi, c, waits in rows[:3]:
            mf = 32 * c.get("mfma32", 0) + 16 * c.get("mfma16", 0)
            valu = 4 * (c.get("valu", 0) + c.get("cvt_pk", 0)) + 8 * c.get("trans", 0)
            lds = c.get("lds_read", 0) + c.get("lds_write", 0) + c.get("lds_permute", 0) + c.get("lds_other", 0)
            print(f"loop at lines {lo}-{hi} ({hi - lo + 1} lines): per iteration and wave")
            print(f"  MFMA {nm} ({c.get('mfma32', 0)} x 32x32, {c.get('mfma16', 0)} x 16x16) = {mf} matrix-pipe cycles")
            print(f"  VALU {c.get('valu', 0)} + packed converts {c.get('cvt_pk', 0)} + transcendentals {c.get('trans', 0)} ~ {valu} issue cycles"
                  f" ({valu / mf:.2f} x the MFMA cycles)")
            print(f"  LDS: {c.get('lds_read', 0)} reads, {c.get('lds_write', 0)} writes, {c.get('lds_permute', 0)} permutes; LDS-DMA / global loads "
                  f"{c.get('lds_dma', 0) + c.get('vmem_load', 0)}; global stores {c.get('vmem_store', 0)}  ({lds} LDS instructions ~ {4 * lds} issue cycles)")
            print(f"  SALU {c.get('salu', 0)}, branches {c.get('branch', 0)}, barriers {c.get('barrier', 0)}, s_waitcnt {c.get('waitcnt', 0)}: "
                  + "; ".join(sorted(set(waits))))
            if os.environ.get("LOOP_MIX_HIST", "1") != "0" and (nm, lo, hi, c, waits) == rows[0]:
                hv = histogram(lines, lo, hi, ("valu", "trans", "cvt_pk"))
                hs = histogram(lines, lo, hi, ("salu",))
                print("  vector opcodes: " + ", ".join(f"{k} {v}" for k, v in hv[:28]))
                print("  scalar opcodes: " + ", ".join(f"{k} {v}" for k, v in hs[:14]))
            issue = valu + 4 * (lds + c.get("salu", 0) + c.get("lds_dma", 0) + c.get("vmem_load", 0) + c.get("vmem_store", 0))
            print(f"  one wave's own issue stream ~ {issue} cycles beside {mf} MFMA cycles: with two waves per SIMD the matrix pipe can be at most "
                  f"{min(1.0, 2 * mf / max(issue + mf, 1)):.0%} busy if nothing overlaps inside a wave, {min(1.0, 2 * mf / max(issue, 2 * mf)):.0%} if everything does\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
