#!/usr/bin/env python
"""Per-kernel PMC counter totals from a rocprofv3 rocpd database.
    python tools/rocpd_pmc.py results.db [--by-grid]
--by-grid: one row group per (kernel, launch grid, duration cluster) -- i.e. per layer shape of a kernel that serves several
layers; launches of one kernel and grid whose durations differ by more than 1.25x are split and numbered by size (size_rank) --
with the number of dispatches and their mean duration next to the per-dispatch counter values."""
import re
import sqlite3
import sys
from collections import defaultdict


def clusters(durs, gap=1.25):
    """Indices of `durs` split into groups of similar duration (sorted; a new group where the next value is > gap x the previous):
    one kernel with one launch grid may still serve layers of different sizes."""
    order = sorted(range(len(durs)), key=lambda i: durs[i])
    out, cur = [], [order[0]]
    for i in order[1:]:
        if durs[i] > gap * durs[cur[-1]]:
            out.append(cur); cur = []
        cur.append(i)
    out.append(cur)
    return out


def main(path, by_grid=False):
    db = sqlite3.connect(path)
    c = db.cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda p: next(x for x in tabs if x.startswith(p))
    kd, ks, pe, pi = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_pmc_event"), t("rocpd_info_pmc")
    names = {r[0]: r[1] for r in c.execute(f"select id, name from {pi}")}
    grid = "d.grid_size_x" if by_grid else "0"
    rows = c.execute(f"select s.kernel_name, {grid}, d.id, d.end - d.start, e.pmc_id, sum(e.value) from {pe} e "
                     f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id "
                     f"group by d.id, e.pmc_id").fetchall()
    disp = defaultdict(dict)          # (kernel, grid) -> dispatch id -> {"dur": ns, counter: value}
    for k, g, did, dur, pid, val in rows:
        short = re.sub(r"\(anonymous namespace\)::", "", k)
        short = re.sub(r"\(.*$", "", short)[:78]
        d = disp[(short, g)].setdefault(did, {"dur": dur})
        d[names.get(pid, str(pid))] = val
    groups = []                       # (kernel, grid, cluster rank, [dispatch dicts])
    for (k, g), dd in disp.items():
        ds = list(dd.values())
        cl = clusters([d["dur"] for d in ds]) if by_grid else [list(range(len(ds)))]
        for rank, idx in enumerate(cl):
            groups.append((k, g, rank if len(cl) > 1 else -1, [ds[i] for i in idx]))
    groups.sort(key=lambda x: -sum(d["dur"] for d in x[3]))
    for k, g, rank, ds in groups:
        n = len(ds)
        head = f"== {k}" + (f"  grid_x={g}" if by_grid else "") + f"  dispatches={n}  mean_us={sum(d['dur'] for d in ds) / n / 1e3:.1f}"
        if rank >= 0:
            head += f"  size_rank={rank}"
        print(head)
        for cn in sorted({c_ for d in ds for c_ in d if c_ != "dur"}):
            v = sum(d.get(cn, 0) for d in ds)
            print(f"   {cn:32s} {v:16.0f}   per-dispatch {v/n:14.0f}")


if __name__ == "__main__":
    main(sys.argv[1], "--by-grid" in sys.argv[2:])
