"""ISA lint for the inline-asm loads of gemm.hip (run on `hipcc -S` output).

hipcc treats the destination of an inline-asm load as valid at the end of the asm statement (it does not model the load), so
nothing may READ or WRITE that register between the load and the inline-asm `s_waitcnt` that covers it -- a compiler-made
copy in that window moves stale data and leaves a register the load overwrites later (memory faults, wrong tiles).
This checks every `ds_read_b128` / `ds_read_b64` / `ds_read_b64_tr_b16` / `global_load_dwordx4 v[..]` that sits inside an ASMSTART/ASMEND pair:
until the next inline-asm s_waitcnt of the matching counter, no other instruction mentions its destination registers.

    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only nemo_amd/csrc/gemm.hip -o gemm.s && python tools/check_asm_loads.py gemm.s
"""
import re
import sys


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def all_vregs(line):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", line):
        out |= regs(tok)
    return out


def main(path):
    lines = open(path).read().split("\n")
    in_asm, pending, bad, checked = False, [], [], 0   # pending: (counter, regs, line_no, text)
    for i, raw in enumerate(lines):
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            if t.endswith(":") and (t.startswith("_Z") or "Lfunc_begin" in t):
                pending = []
            continue
        op = t.split()[0]
        if in_asm and op.startswith("s_waitcnt"):
            # a counted wait retires the OLDEST loads of its counter and leaves the newest N pending (in-order return)
            for cname, key in (("lgkmcnt", "lgkm"), ("vmcnt", "vm")):
                m = re.search(cname + r"\((\d+)\)", t)
                if m:
                    n = int(m.group(1))
                    mine = [p for p in pending if p[0] == key]
                    keep = mine[len(mine) - n:] if n < len(mine) else mine
                    pending = [p for p in pending if p[0] != key] + keep
            continue
        if in_asm and op in ("ds_read_b128", "ds_read_b64", "ds_read_b32", "ds_read_b64_tr_b16", "global_load_dwordx4"):
            if op == "global_load_dwordx4" and "lds" in t:
                continue
            dst = regs(t.split()[1].rstrip(","))
            srcs = all_vregs(t.split(",", 1)[1]) if "," in t else set()
            for cnt, d0, ln, txt in pending:  # an address register that is itself the destination of an un-waited load
                if srcs & d0:
                    bad.append((i + 1, t, ln, txt))
            pending.append(("lgkm" if op.startswith("ds_") else "vm", dst, i + 1, t))
            checked += 1
            continue
        if op in ("s_endpgm",):
            pending = []
            continue
        used = all_vregs(t)
        for cnt, dst, ln, txt in pending:
            if used & dst:
                bad.append((i + 1, t, ln, txt))
    print(f"[check_asm_loads] {checked} inline-asm loads checked, {len(bad)} violations")
    for b in bad[:20]:
        print("  line %d `%s` touches the destination of the un-waited asm load at line %d `%s`" % b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
