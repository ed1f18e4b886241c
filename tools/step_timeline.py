#!/usr/bin/env python
"""Timeline of ONE training step out of a rocprofv3 kernel trace (rocpd sqlite): every dispatch in start order with its queue,
start offset, duration, the idle gap in front of it on its own queue, and how many other dispatches were in flight when it started.
A step starts at the image cast kernel of the forward (`cast_kernel` / `image_to_cl`), one per forward.
    python tools/step_timeline.py results.db [--step -2] [--brief]
Footer: union busy time of the step, time with >= 2 dispatches in flight, idle time, per-queue busy time, launches per step."""
import argparse
import re
import sqlite3


def short(name):
    s = re.sub(r"\(anonymous namespace\)::", "", name)
    s = re.sub(r"^void ", "", s)
    s = re.sub(r"\(.*$", "", s)
    s = s.replace("at::native::", "")
    return s[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--step", type=int, default=-2, help="index of the step to print (negative: from the end)")
    ap.add_argument("--brief", action="store_true", help="footer only")
    ap.add_argument("--marker", default="cast_kernel,image_to_cl", help="kernel-name substrings that open a step")
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    c = db.cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    rows = c.execute(f"select d.start, d.end, d.queue_id, d.stream_id, s.kernel_name, d.grid_size_x, d.grid_size_y, d.workgroup_size_x "
                     f"from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    marks = [m for m in a.marker.split(",") if m]
    starts = [i for i, r in enumerate(rows) if any(m in r[4] for m in marks)]
    if len(starts) < 3:
        raise SystemExit(f"only {len(starts)} step markers in the trace")
    k = a.step if a.step >= 0 else len(starts) + a.step
    lo, hi = starts[k], starts[k + 1] if k + 1 < len(starts) else len(rows)
    step = rows[lo:hi]
    t0 = step[0][0]
    t_end = max(r[1] for r in step)
    last_end = {}
    if not a.brief:
        print(f"# step {k} of {len(starts)}: {len(step)} dispatches, span {(t_end - t0) / 1e3:.1f} us")
        print(f"{'t_us':>9s} {'dur_us':>8s} {'gap_us':>7s} {'q':>2s} {'inflight':>8s}  kernel [grid/wg]")
    for i, (s, e, q, st, name, gx, gy, wx) in enumerate(step):
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = max(e, last_end.get(q, 0))
        infl = sum(1 for r in step[max(0, i - 40):i] if r[1] > s)
        if not a.brief:
            print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f} {q:2d} {infl:8d}  {short(name)} [{gx // max(wx, 1)}x{gy}]")
    # union / overlap by sweep
    ev = []
    for s, e, *_ in step:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    busy = over = 0
    depth, prev = 0, ev[0][0]
    for t, d in ev:
        if depth >= 1:
            busy += t - prev
        if depth >= 2:
            over += t - prev
        depth += d
        prev = t
    span = t_end - t0
    perq = {}
    for s, e, q, *_ in step:
        perq[q] = perq.get(q, 0) + (e - s)
    tiny = [r for r in step if r[1] - r[0] < 12000]
    print(f"# step span {span / 1e6:.3f} ms | busy (union) {busy / 1e6:.3f} | >=2 in flight {over / 1e6:.3f} | idle {(span - busy) / 1e6:.3f} | "
          f"sum of durations {sum(r[1] - r[0] for r in step) / 1e6:.3f}")
    print("# per queue busy ms: " + ", ".join(f"q{q}: {v / 1e6:.3f}" for q, v in sorted(perq.items())))
    print(f"# dispatches {len(step)}, of which < 12 us: {len(tiny)} ({sum(r[1] - r[0] for r in tiny) / 1e6:.3f} ms)")
    # next-step start relative to this step's end (host-side gap between steps)
    if k + 1 < len(starts):
        print(f"# next step's first dispatch starts {(rows[starts[k + 1]][0] - t_end) / 1e3:.1f} us after this step's last end")


if __name__ == "__main__":
    main()
