"""Per-stream busy time and inter-kernel gaps from a rocprofv3 kernel trace (rocpd sqlite): how much of the step is a stream
sitting idle between two of its own launches (launch gaps) rather than running kernels.
    python tools/stream_gaps.py <trace.db> [last-ms-window]"""
import sqlite3
import sys


def main(db_path, window_ms=0.0):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
    rows = list(cur.execute(f"select {qcol}, start, end from {kd} order by start"))
    if window_ms:  # only the last `window_ms` of the trace (the timed steps; model construction etc. comes before)
        t_end = max(r[2] for r in rows)
        rows = [r for r in rows if r[1] >= t_end - window_ms * 1e6]
    t_first, t_last = rows[0][1], max(r[2] for r in rows)
    by = {}
    for q, s, e in rows:
        by.setdefault(q, []).append((s, e))
    print(f"{len(rows)} dispatches over {(t_last - t_first) / 1e6:.2f} ms, grouped by {qcol}")
    for q, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e in v)
        gaps = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
        pos = [g for g in gaps if g > 0]
        small = [g for g in pos if g < 50_000]  # < 50 us: launch-to-launch gaps, not waits for another phase
        if q == max(by, key=lambda k: len(by[k])):
            hist = {}
            for g in small:
                hist[int(g // 2000) * 2] = hist.get(int(g // 2000) * 2, 0) + 1
            print("   main-stream gap histogram (us bucket: count):", dict(sorted(hist.items())))
        print(f"  {qcol} {q}: {len(v)} kernels, busy {busy / 1e6:.2f} ms, span {(v[-1][1] - v[0][0]) / 1e6:.2f} ms, "
              f"gaps<50us: n={len(small)} sum {sum(small) / 1e6:.2f} ms median {sorted(small)[len(small) // 2] / 1e3 if small else 0:.1f} us, "
              f"gaps>=50us: n={len(pos) - len(small)} sum {(sum(pos) - sum(small)) / 1e6:.2f} ms")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.0)
