"""ISA lint for the inline-asm loads of gemm.hip / attention.hip (run on `hipcc -S` output).

hipcc treats the destination of an inline-asm load as valid at the end of the asm statement (it does not model the load), so
nothing may READ or WRITE that register between the load and the inline-asm `s_waitcnt` that covers it -- a compiler-made
copy in that window moves stale data and leaves a register the load overwrites later (memory faults, wrong tiles).
This checks every `ds_read_b128` / `ds_read_b64` / `ds_read_b64_tr_b16` / `global_load_dwordx4 v[..]` that sits inside an ASMSTART/ASMEND pair:
until an inline-asm s_waitcnt of the matching counter retires it, no other instruction mentions its destination registers.
A counted `lgkmcnt(N)` retires the oldest LDS reads and leaves the newest N pending (in-order return).
The walk follows the control flow (basic blocks, both sides of a conditional branch): the compiler places blocks in any order, so
the text after a load is not necessarily what executes after it.

    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only nemo_amd/csrc/gemm.hip -o gemm.s && python tools/check_asm_loads.py gemm.s
"""
import re
import sys

LOADS = ("ds_read_b128", "ds_read_b64", "ds_read_b32", "ds_read_b64_tr_b16", "global_load_dwordx4")


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return frozenset(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return frozenset({int(m.group(1))}) if m else frozenset()


def all_vregs(line):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", line):
        out |= regs(tok)
    return out


def parse_functions(lines):
    """-> list of functions, each {label: [(line_no, text, in_asm)]} with an ordered label list"""
    funcs, cur, order, label, in_asm = [], None, None, None, False
    for i, raw in enumerate(lines):
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        head = t.split(";", 1)[0].strip()   # (labels carry trailing comments: `.LBB1_2:   ; in Loop ...`)
        if head.endswith(":") and not t.startswith((";", "//")):
            name = head[:-1]
            if name.startswith("_Z") or re.match(r"^[A-Za-z_][\w.$]*$", name) and not name.startswith(".L"):
                cur, order = {}, []
                funcs.append((cur, order))
                label = name
                cur[label] = []
                order.append(label)
                continue
            if cur is not None and name.startswith(".LBB"):
                label = name
                cur[label] = []
                order.append(label)
                continue
        if cur is None or not t or t.startswith((";", ".", "//")):
            continue
        cur[label].append((i + 1, t, in_asm))
    return funcs


def check_function(blocks, order, bad):
    checked = 0
    nxt = {order[i]: (order[i + 1] if i + 1 < len(order) else None) for i in range(len(order))}
    seen = set()
    stack = [(order[0], ())]
    while stack:
        label, pending = stack.pop()
        while label is not None:
            key = (label, frozenset((p[0], p[1]) for p in pending))
            if key in seen:
                break
            seen.add(key)
            pending = list(pending)
            stop = False
            for ln, t, in_asm in blocks[label]:
                op = t.split()[0]
                if in_asm and op.startswith("s_waitcnt"):
                    for cname, kind in (("lgkmcnt", "lgkm"), ("vmcnt", "vm")):
                        m = re.search(cname + r"\((\d+)\)", t)
                        if m:
                            # lgkmcnt: every LDS read of these loops is an asm load the walk has seen, the count is exact.
                            # vmcnt also counts the LDS-DMA and the stores in flight, which are not tracked here: the kernels only
                            # wait for asm global loads with counts that cover them (tile-ring DMA issued AFTER them), so any
                            # inline vmcnt wait retires them all (the rule of the previous version of this lint)
                            n = int(m.group(1)) if kind == "lgkm" else 0
                            mine = [p for p in pending if p[0] == kind]
                            keep = mine[len(mine) - n:] if n < len(mine) else mine
                            pending = [p for p in pending if p[0] != kind] + keep
                    continue
                if in_asm and op == "global_load_dwordx4":
                    continue  # (global asm loads: linear pass below)
                if in_asm and op in LOADS:
                    dst = regs(t.split()[1].rstrip(","))
                    srcs = all_vregs(t.split(",", 1)[1]) if "," in t else set()
                    for kind, d0, l0, t0 in pending:  # an address register that is itself the destination of an un-waited load
                        if srcs & d0:
                            bad.add((ln, t, l0, t0))
                    pending.append(("lgkm" if op.startswith("ds_") else "vm", dst, ln, t))
                    checked += 1
                    continue
                if op == "s_endpgm":
                    stop = True
                    break
                used = all_vregs(t)
                for kind, dst, l0, t0 in pending:
                    if used & dst:
                        bad.add((ln, t, l0, t0))
                if op == "s_branch":
                    label = t.split()[1]
                    stop = True
                    stack.append((label, tuple(pending)))
                    break
                if op.startswith("s_cbranch"):
                    tgt = t.split()[1]
                    if tgt in blocks:
                        stack.append((tgt, tuple(pending)))
                if op.startswith("s_setpc") or op.startswith("s_swappc"):
                    stop = True
                    break
            if stop:
                break
            label = nxt.get(label)
            pending = tuple(pending)
            if label is not None and not pending and (label, frozenset()) in seen:
                break
    return checked


def check_global_loads_linear(lines, bad):
    """inline-asm `global_load_dwordx4 v[..]` (aux_in prefetch of the persistent GEMM): covered by ONE of several counted vmcnt
    waits selected by correlated uniform branches (a switch) -- a path-insensitive walk would take the path that skips them all, so
    these are checked in text order: until the next inline-asm vmcnt wait nothing mentions the destination (the rule this lint
    started with)"""
    in_asm, pending, checked = False, [], 0
    for i, raw in enumerate(lines):
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        head = t.split(";", 1)[0].strip()
        if not t or t.startswith((";", ".", "//")) or head.endswith(":"):
            if head.endswith(":") and head.startswith("_Z"):
                pending = []
            continue
        op = t.split()[0]
        if in_asm and op.startswith("s_waitcnt"):
            if "vmcnt" in t:
                pending = []
            continue
        if in_asm and op == "global_load_dwordx4" and "lds" not in t:
            pending.append((regs(t.split()[1].rstrip(",")), i + 1, t))
            checked += 1
            continue
        if op == "s_endpgm":
            pending = []
            continue
        used = all_vregs(t)
        for dst, l0, t0 in pending:
            if used & dst:
                bad.add((i + 1, t, l0, t0))
    return checked


def main(path):
    lines = open(path).read().split("\n")
    bad, checked = set(), 0
    checked += check_global_loads_linear(lines, bad)
    for blocks, order in parse_functions(lines):
        if not order:
            continue
        checked += check_function(blocks, order, bad)
    bad = sorted(bad)
    print(f"[check_asm_loads] {checked} inline-asm load sites visited (control-flow walk), {len(bad)} violations")
    for b in bad[:20]:
        print("  line %d `%s` touches the destination of the un-waited asm load at line %d `%s`" % b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
