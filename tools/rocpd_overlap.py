#!/usr/bin/env python
"""Concurrency analysis of a rocprofv3 kernel trace (rocpd sqlite): over the last `frac` of the trace (steady-state steps) --
wall span, time with >= 1 / >= 2 kernels in flight, sum of kernel durations by class (MFMA-bound conv / weight-gradient kernels vs
HBM-bound normalisation-type kernels), and how much of the HBM-bound kernels' time ran under an MFMA kernel.
    python tools/rocpd_overlap.py results.db [frac=0.5]"""
import re
import sqlite3
import sys

MFMA = re.compile(r"igemm|conv_c1|wgrad")


def main(path, frac=0.5):
    db = sqlite3.connect(path)
    c = db.cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    rows = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    lo = t1 - (t1 - t0) * frac
    rows = [r for r in rows if r[1] >= lo]
    ev = []
    for name, s, e in rows:
        m = 1 if MFMA.search(name) else 0
        ev.append((s, 1, m)); ev.append((e, -1, m))
    ev.sort()
    n = [0, 0]          # in flight: [hbm-type, mfma-type]
    last = ev[0][0]
    busy1 = busy2 = hbm_under_mfma = hbm_time = mfma_time = both_mfma = 0
    for t, d, m in ev:
        dt = t - last
        tot = n[0] + n[1]
        if tot >= 1: busy1 += dt
        if tot >= 2: busy2 += dt
        if n[0] >= 1: hbm_time += dt
        if n[1] >= 1: mfma_time += dt
        if n[0] >= 1 and n[1] >= 1: hbm_under_mfma += dt
        if n[1] >= 2: both_mfma += dt
        n[m] += d
        last = t
    span = ev[-1][0] - ev[0][0]
    sums = {}
    for name, s, e in rows:
        k = "mfma" if MFMA.search(name) else "hbm"
        sums[k] = sums.get(k, 0) + (e - s)
    ms = lambda x: x / 1e6
    print(f"window {ms(span):.2f} ms, {len(rows)} dispatches: >=1 kernel in flight {ms(busy1):.2f} ms, >=2 {ms(busy2):.2f} ms")
    print(f"  MFMA-type kernels: sum of durations {ms(sums.get('mfma', 0)):.2f} ms, time with >=1 in flight {ms(mfma_time):.2f} ms, >=2 in flight {ms(both_mfma):.2f} ms")
    print(f"  HBM-type kernels : sum of durations {ms(sums.get('hbm', 0)):.2f} ms, time with >=1 in flight {ms(hbm_time):.2f} ms, "
          f"of which under an MFMA kernel {ms(hbm_under_mfma):.2f} ms")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)
