#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace: per-kernel calls / total / average / share,
the same table `rocprofv3 --stats` prints in csv mode.  Usage: python tools/rocpd_stats.py results.db [kernel-substring]
With a kernel substring, the individual dispatches of that kernel are listed grouped by grid size (one group per
layer shape): the group launched back to back by bench.py's roofline section is the one its `roofline.launch_ms`
has to agree with."""
import re
import sqlite3
import sys


def main(path, steps=None):
    db = sqlite3.connect(path)
    c = db.cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    rows = c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                     f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    span = c.execute(f"select min(start), max(end) from {kd}").fetchone()
    print(f"# kernels: {sum(r[1] for r in rows)} dispatches, busy {total/1e6:.2f} ms, span {(span[1]-span[0])/1e6:.2f} ms")
    print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for name, n, tot, mn, mx in rows:
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*$", "", short)[:90]
        print(f"{short:90s} {n:7d} {tot/1e6:10.3f} {tot/n/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*tot/total:6.2f}")


    if steps:
        pat = steps
        drows = c.execute(f"select d.grid_size_x, d.grid_size_y, d.end-d.start from {kd} d join {ks} s on d.kernel_id = s.id "
                          f"where s.kernel_name like ? order by d.start", (f"%{pat}%",)).fetchall()
        groups = {}
        for gx, gy, dur in drows:
            groups.setdefault((gx, gy), []).append(dur / 1e3)
        print(f"\n# dispatches of *{pat}* by grid (threads x, y): count, avg / min / max us")
        for (gx, gy), ds in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
            print(f"  grid ({gx:8d},{gy:4d}) n={len(ds):4d}  avg {sum(ds)/len(ds):9.2f}  min {min(ds):9.2f}  max {max(ds):9.2f}")
            top = sorted(ds, reverse=True)[:8]
            print("      longest dispatches (us): " + ", ".join(f"{d:.1f}" for d in top)
                  + f"   -> mean of the 5 longest {sum(top[:5])/max(1, len(top[:5])):.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
