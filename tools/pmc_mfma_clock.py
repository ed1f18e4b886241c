"""profiles/r04_pmc_mfma_clock.json from a `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
SQ_BUSY_CYCLES` pass over bench.py (tools/gpu_r5_final.sh): for every MFMA kernel group of the step

  clock_ghz      = GRBM_GUI_ACTIVE per dispatch / 8 XCDs (the counter is summed over the XCDs) / launch duration
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8): share of the matrix pipes' CYCLES
                   that carry an MFMA (the counter advances 32 per v_mfma_f32_32x32x16_f16 and SIMD, checked against SQ_INSTS_MFMA)
  frac_of_peak_at_measured_clock = mfma_busy_frac; x clock_ghz / 2.4 = fraction of the 2.5 PFLOP/s datasheet peak

The launch DURATION comes from a pass without GRBM / SQ counters (the FETCH_SIZE pass of the same script): rocprofv3 reports
exactly 8x the duration for every kernel when GRBM_GUI_ACTIVE is collected (memory-bound kernels come out at the nominal
2.2-2.5 GHz with the unperturbed duration, which is the check that this reading is right); ratios inside one pass are unaffected.

usage: python tools/pmc_mfma_clock.py gpurun_out/<tag>/pmc_mfma.txt gpurun_out/<tag>/pmc_fetch.txt profiles/r04_pmc_mfma_clock.json
"""
import json
import re
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import short  # noqa: E402


def parse(src):
    rows, key = {}, None
    for line in open(src):
        m = re.match(r"== (\S+)\s+grid_x=(\d+)\s+dispatches=(\d+)\s+mean_us=([\d.]+)(?:\s+size_rank=(\d+))?", line)
        if m:
            key = (re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", m.group(1)), int(m.group(2)), int(m.group(5) or -1))
            rows[key] = {"dispatches": int(m.group(3)), "mean_us": float(m.group(4))}
            continue
        m = re.match(r"\s+(\w+)\s+\d+\s+per-dispatch\s+(\d+)", line)
        if m and key:
            rows[key][m.group(1)] = int(m.group(2))
    return rows



def so_sha256():
    """sha256 of the liblnn_hip.so these counters were collected with: bench.py quotes the file only for the same binary."""
    import hashlib
    import os
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lifelong-nnunet_amd", "csrc", "liblnn_hip.so")
    return hashlib.sha256(open(so, "rb").read()).hexdigest()


def main(src, src_clock, dst):
    rows, clk = parse(src), parse(src_clock)
    out = []
    for (name, grid, rank), r in rows.items():
        if r.get("SQ_INSTS_MFMA", 0) == 0 or "GRBM_GUI_ACTIVE" not in r or r["mean_us"] < 40:
            continue
        cyc = r["GRBM_GUI_ACTIVE"] / 8.0
        busy = r["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
        c = clk.get((name, grid, rank))
        if not c:
            continue
        ghz = cyc / c["mean_us"] / 1e3
        r["mean_us"] = c["mean_us"]
        out.append({"kernel": short(name), "grid_x": grid, "size_rank": rank, "dispatches": r["dispatches"], "mean_us": r["mean_us"],
                    "clock_ghz": round(ghz, 3), "mfma_busy_frac_in_cycles": round(busy, 3),
                    "busy_cycles_per_mfma_inst": round(r["SQ_VALU_MFMA_BUSY_CYCLES"] / r["SQ_INSTS_MFMA"], 2),
                    "frac_of_2p5_pflops": round(busy * ghz / 2.4, 3)})
    out.sort(key=lambda r: -r["mean_us"] * r["dispatches"])
    pick = lambda k, g: max((r for r in out if r["kernel"] == k and r["grid_x"] == g), key=lambda r: r["mean_us"], default=None)
    fam = {"fwd": pick("conv_s1_v9<4,1,2,epi=1>", 131072), "dgrad": pick("conv_s1_v9<2,2,2,epi=0>", 131072),
           "wgrad": pick("wgrad_s1_v5", 131072)}     # the launches of conv_blocks_localization.4.0 (bench.py's roofline block)
    json.dump({"note": __doc__.split("usage")[0].strip(), "so_sha256": so_sha256(), "families": fam, "kernels": out},
              open(dst, "w"), indent=1)
    for r in out[:24]:
        print(f"{r['kernel']:36s} g={r['grid_x']:8d} {r['mean_us']:8.1f} us  {r['clock_ghz']:.2f} GHz  busy {r['mfma_busy_frac_in_cycles']:.2f}"
              f"  ({r['busy_cycles_per_mfma_inst']} cyc/inst)  -> {r['frac_of_2p5_pflops']:.2f} of 2.5 PF")


if __name__ == "__main__":
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    main(*sys.argv[1:4])
