#!/usr/bin/env python
"""Markdown table of the C2 step by kernel group from the committed evidence: launches and time per step (serialised trace),
HBM bytes and TB/s per launch (PMC traffic pass), matrix-pipe busy share and shader clock (PMC MFMA pass).
    python tools/step_table.py > /tmp/table.md        (inputs: profiles/r04_pmc_traffic.json, r04_pmc_mfma_clock.json)"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tr = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")))["all_kernels_over_40us"]
    ck = {(k["kernel"], k["grid_x"], k["size_rank"]): k
          for k in json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_mfma_clock.json")))["kernels"]}
    rows = sorted(tr, key=lambda r: -r["mean_us"] * r["launches_per_step"])
    tot = sum(r["mean_us"] * r["launches_per_step"] for r in rows)
    print("| kernel (launch-size cluster) | launches / step | µs / launch | ms / step | HBM GB / launch | TB/s | pipes busy | clock GHz |")
    print("|---|---|---|---|---|---|---|---|")
    for r in rows[:28]:
        c = ck.get((r["kernel"], r["grid_x"], r["size_rank"]))
        busy = f"{c['mfma_busy_frac_in_cycles']:.2f}" if c else "–"
        ghz = f"{c['clock_ghz']:.2f}" if c else "–"
        print(f"| `{r['kernel']}` | {r['launches_per_step']:.0f} | {r['mean_us']:.0f} | {r['mean_us'] * r['launches_per_step'] / 1e3:.2f} | "
              f"{r['hbm_bytes_per_launch_corrected'] / 1e9:.2f} | {r['hbm_tb_per_s']:.1f} | {busy} | {ghz} |")
    print(f"\n(kernels above 40 µs: {tot / 1e3:.1f} ms per step in this serialised PMC pass)")


if __name__ == "__main__":
    main()
