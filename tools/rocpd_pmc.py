#!/usr/bin/env python
"""Per-kernel PMC counter totals from a rocprofv3 rocpd database.
    python tools/rocpd_pmc.py results.db [--by-grid]
--by-grid: one row group per (kernel, launch grid) -- i.e. per layer shape of a kernel that serves several layers -- with the
number of dispatches and their mean duration next to the per-dispatch counter values."""
import re
import sqlite3
import sys
from collections import defaultdict


def main(path, by_grid=False):
    db = sqlite3.connect(path)
    c = db.cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda p: next(x for x in tabs if x.startswith(p))
    kd, ks, pe, pi = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_pmc_event"), t("rocpd_info_pmc")
    names = {r[0]: r[1] for r in c.execute(f"select id, name from {pi}")}
    grid = ", d.grid_size_x" if by_grid else ", 0"
    rows = c.execute(f"select s.kernel_name{grid}, e.pmc_id, sum(e.value), count(distinct d.id), sum(d.end-d.start) from {pe} e "
                     f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id "
                     f"group by s.kernel_name{grid}, e.pmc_id").fetchall()
    agg = defaultdict(dict)
    meta = {}
    for k, g, pid, val, n, dur in rows:
        short = re.sub(r"\(anonymous namespace\)::", "", k)
        short = re.sub(r"\(.*$", "", short)[:78]
        key = (short, g)
        agg[key][names.get(pid, str(pid))] = val
        meta[key] = (n, dur)
    order = sorted(agg, key=lambda k: -max(agg[k].values()))
    for key in order:
        n, dur = meta[key]
        k, g = key
        head = f"== {k}" + (f"  grid_x={g}" if by_grid else "") + f"  dispatches={n}  mean_us={dur / n / 1e3:.1f}"
        print(head)
        for cn, v in sorted(agg[key].items()):
            print(f"   {cn:32s} {v:16.0f}   per-dispatch {v/n:14.0f}")


if __name__ == "__main__":
    main(sys.argv[1], "--by-grid" in sys.argv[2:])
