"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share -> markdown + csv."""
import re
import sqlite3
import sys


def main(db_path, out_prefix=None, skip_first_frac=0.0):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    scol = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else "name")
    q = f"select s.{name_col}, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"
    rows = list(cur.execute(q))
    agg = {}
    for name, st, en in rows:
        name = re.sub(r"\(.*", "", name)
        name = re.sub(r"^void ", "", name)
        a = agg.setdefault(name, [0, 0])
        a[0] += 1
        a[1] += en - st
    total = sum(v[1] for v in agg.values())
    lines = ["| kernel | calls | total ms | avg us | % |", "|---|---:|---:|---:|---:|"]
    csv = ["kernel,calls,total_ns,avg_ns,percent"]
    for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| `{name[:90]}` | {n} | {t/1e6:.3f} | {t/n/1e3:.1f} | {100*t/total:.2f} |")
        csv.append(f"\"{name}\",{n},{t},{t/n:.0f},{100*t/total:.3f}")
    lines.append(f"| **total GPU kernel time** | {sum(v[0] for v in agg.values())} | {total/1e6:.3f} | | 100 |")
    text = "\n".join(lines)
    print(text)
    if out_prefix:
        open(out_prefix + ".md", "w").write(text + "\n")
        open(out_prefix + ".csv", "w").write("\n".join(csv) + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
