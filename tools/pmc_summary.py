"""Per-kernel average of a rocprofv3 PMC counter (rocpd sqlite): kernel, launches, mean counter value, mean duration."""
import re
import sqlite3
import sys


def main(db_path, counter=None):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    def tab(prefix): return [t for t in tabs if t.startswith(prefix)][0]
    kd, ks, pe, ip = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
    scol = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scol else "display_name"
    pcols = [r[1] for r in cur.execute(f"pragma table_info({pe})")]
    q = (f"select s.{name_col}, i.name, p.value, d.end - d.start from {pe} p join {ip} i on p.pmc_id = i.id "
         f"join {kd} d on p.event_id = d.event_id join {ks} s on d.kernel_id = s.id")
    agg = {}
    for name, cname, val, dur in cur.execute(q):
        if counter and cname != counter:
            continue
        name = re.sub(r"\(.*", "", name)
        a = agg.setdefault((name, cname), [0, 0.0, 0])
        a[0] += 1; a[1] += val; a[2] += dur
    print("| kernel | counter | launches | mean value | mean us |\n|---|---|---:|---:|---:|")
    for (name, cname), (n, v, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{name[:80]}` | {cname} | {n} | {v / n:.1f} | {d / n / 1e3:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
