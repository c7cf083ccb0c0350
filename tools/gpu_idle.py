"""GPU idle time inside the timed steps of a rocprofv3 kernel trace (rocpd sqlite): union of the kernel intervals (all streams)
against the span from the first to the last kernel of the window -- tells whether the host's launch rate ever starves the GPU."""
import sqlite3
import sys


def main(db_path, skip_frac=0.4):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    scol = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scol else "display_name"
    rows = sorted(cur.execute(f"select d.start, d.end, s.{name_col} from {kd} d join {ks} s on d.kernel_id = s.id"))
    rows = rows[int(len(rows) * skip_frac):]  # drop set-up and warm-up
    span = rows[-1][1] - rows[0][0]
    busy, cs, ce = 0, rows[0][0], rows[0][1]
    gaps = []
    last_name = rows[0][2]
    where = []
    for s, e, name in rows[1:]:
        if s > ce:
            busy += ce - cs
            gaps.append(s - ce)
            where.append((s - ce, last_name[:48], name[:48]))
            cs, ce = s, e
            last_name = name
        else:
            if e > ce:
                ce, last_name = e, name
    busy += ce - cs
    gaps.sort(reverse=True)
    print(f"window {span/1e6:.2f} ms, {len(rows)} kernels, busy (any stream) {busy/1e6:.2f} ms = {100*busy/span:.1f} %, idle {100*(1-busy/span):.1f} %")
    print("gaps: n =", len(gaps), " >20us:", sum(1 for g in gaps if g > 20000), " >5us:", sum(1 for g in gaps if g > 5000),
          " largest (us):", [round(g / 1e3, 1) for g in gaps[:8]], " sum of gaps <5us (ms):", round(sum(g for g in gaps if g <= 5000) / 1e6, 2))


    where.sort(reverse=True)
    for g, a, b in where[:12]:
        print(f"  gap {g/1e3:8.1f} us   after {a}   before {b}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.4)
