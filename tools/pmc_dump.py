"""Dump a rocprofv3 PMC run (rocpd sqlite) as JSON: per kernel and counter -> dispatches, samples, sum of all samples, mean
duration.  `mean per dispatch` = sum / dispatches; `mean per sample` = sum / samples (rocprofv3 may emit several samples per
dispatch: which of the two is the per-launch value is settled by calibration on kernels with known compulsory traffic,
see tools/pmc_assemble.py)."""
import json
import re
import sqlite3
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    def tab(prefix): return [t for t in tabs if t.startswith(prefix)][0]
    kd, ks, pe, ip = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
    scol = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scol else "display_name"
    q = (f"select s.{name_col}, i.name, p.value, d.end - d.start, d.event_id from {pe} p join {ip} i on p.pmc_id = i.id "
         f"join {kd} d on p.event_id = d.event_id join {ks} s on d.kernel_id = s.id")
    agg = {}
    for name, cname, val, dur, ev in cur.execute(q):
        name = re.sub(r"\(.*", "", name)
        name = re.sub(r"^void ", "", name)
        a = agg.setdefault(name, {}).setdefault(cname, {"samples": 0, "sum": 0.0, "dur_ns": 0, "events": set()})
        a["samples"] += 1
        a["sum"] += val
        if ev not in a["events"]:
            a["events"].add(ev)
            a["dur_ns"] += dur
    out = {}
    for name, cs in agg.items():
        out[name] = {c: {"dispatches": len(a["events"]), "samples": a["samples"], "sum": a["sum"],
                         "mean_us": a["dur_ns"] / max(1, len(a["events"])) / 1e3} for c, a in cs.items()}
    json.dump(out, open(out_path, "w"), indent=0)
    print(f"{out_path}: {len(out)} kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
