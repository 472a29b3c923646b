#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace [--stats] [--pmc ...]`):
per kernel the number of dispatches, average / min / max duration, grid size, and the per-dispatch average of every
collected counter.  Usage: rocpd_summary.py <results.db> [min_grid]   (text to stdout; the files under profiles/ are its output)."""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    min_grid = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    cur = db.cursor()
    names = {r[0]: r[1] for r in cur.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
    rows = cur.execute("select kernel_id, start, end, grid_size_x, workgroup_size_x, event_id, dispatch_id from rocpd_kernel_dispatch").fetchall()
    pmc = defaultdict(dict)
    try:
        for ev, cname, val in cur.execute("select e.event_id, i.name, e.value from rocpd_pmc_event e join rocpd_info_pmc i on e.pmc_id = i.id"):
            pmc[ev][cname] = pmc[ev].get(cname, 0.0) + val
    except sqlite3.Error:
        pass
    stat = defaultdict(lambda: {"n": 0, "sum": 0, "min": 1 << 62, "max": 0, "grid": set(), "pmc": defaultdict(float), "npmc": 0})
    for kid, s, e, gx, wx, ev, did in rows:
        if gx < min_grid:
            continue
        k = stat[(names.get(kid, str(kid)), gx // max(wx, 1))]
        d = e - s
        k["n"] += 1
        k["sum"] += d
        k["min"] = min(k["min"], d)
        k["max"] = max(k["max"], d)
        if ev in pmc:
            k["npmc"] += 1
            for c, v in pmc[ev].items():
                k["pmc"][c] += v
    total = sum(k["sum"] for k in stat.values()) or 1
    print(f"{'calls':>7} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6} {'workgroups':>10}  kernel")
    for (name, wgs), k in sorted(stat.items(), key=lambda kv: -kv[1]["sum"]):
        line = f"{k['n']:7d} {k['sum'] / k['n'] / 1e3:10.2f} {k['min'] / 1e3:10.2f} {k['max'] / 1e3:10.2f} {100.0 * k['sum'] / total:6.2f} {wgs:10d}  {name[:150]}"
        print(line)
        if k["npmc"]:
            print("        counters per dispatch: " + ", ".join(f"{c} = {v / k['npmc']:.1f}" for c, v in sorted(k["pmc"].items())))


if __name__ == "__main__":
    main()
