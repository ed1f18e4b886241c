#!/usr/bin/env python
"""Per-kernel PMC counter totals from a rocprofv3 rocpd database.  Usage: python tools/rocpd_pmc.py results.db"""
import re
import sqlite3
import sys
from collections import defaultdict


def main(path):
    db = sqlite3.connect(path)
    c = db.cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda p: next(x for x in tabs if x.startswith(p))
    kd, ks, pe, pi = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_pmc_event"), t("rocpd_info_pmc")
    cols = [r[1] for r in c.execute(f"pragma table_info({pe})")]
    names = {r[0]: r[1] for r in c.execute(f"select id, name from {pi}")}
    rows = c.execute(f"select s.kernel_name, e.pmc_id, sum(e.value), count(distinct d.id), sum(d.end-d.start) from {pe} e "
                     f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by s.kernel_name, e.pmc_id").fetchall()
    agg = defaultdict(dict)
    meta = {}
    for k, pid, val, n, dur in rows:
        short = re.sub(r"\(anonymous namespace\)::", "", k)[:70]
        agg[short][names.get(pid, str(pid))] = val
        meta[short] = (n, dur)
    for k, d in agg.items():
        n, dur = meta[k]
        print(f"== {k}  dispatches={n}")
        for cn, v in sorted(d.items()):
            print(f"   {cn:32s} {v:16.0f}   per-dispatch {v/n:14.0f}")


if __name__ == "__main__":
    main(sys.argv[1])
