"""Where are the idle gaps of the main stream?  For every gap >= MIN us between two consecutive dispatches of the busiest stream of a
rocprofv3 kernel trace (rocpd sqlite): the kernel before, the kernel after, the gap, and what the other streams ran meanwhile.
    python tools/gap_sites.py <trace.db> [window-ms] [min-us]"""
import re
import sqlite3
import sys


def main(path, window_ms=200.0, min_us=30.0):
    db = sqlite3.connect(path); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    qcol = "stream_id" if "stream_id" in cols else "queue_id"
    scol = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    ncol = "kernel_name" if "kernel_name" in scol else "display_name"
    rows = list(cur.execute(f"select d.{qcol}, d.start, d.end, s.{ncol} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    t_end = max(r[2] for r in rows)
    rows = [(q, s, e, re.sub(r"\(.*", "", n)[:44]) for q, s, e, n in rows if s >= t_end - window_ms * 1e6]
    cnt = {}
    for r in rows:
        cnt[r[0]] = cnt.get(r[0], 0) + 1
    main_q = max(cnt, key=cnt.get)
    m = [r for r in rows if r[0] == main_q]
    others = [r for r in rows if r[0] != main_q]
    agg = {}
    for a, b in zip(m, m[1:]):
        gap = (b[1] - a[2]) / 1e3
        if gap < min_us:
            continue
        co = [o for o in others if o[1] < b[1] and o[2] > a[2]]
        key = (a[3], b[3])
        g = agg.setdefault(key, [0, 0.0, 0])
        g[0] += 1; g[1] += gap; g[2] += 1 if co else 0
    print(f"main stream {main_q}: {len(m)} dispatches; gaps >= {min_us} us by (kernel before -> kernel after): count, total us, with a side-stream kernel running")
    for (ka, kb), (c, t, co) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
        print(f"  {ka:44s} -> {kb:44s} n={c:3d} total {t:8.1f} us  side-busy {co}")


if __name__ == "__main__":
    main(sys.argv[1], *(float(x) for x in sys.argv[2:4]))
