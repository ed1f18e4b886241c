"""Turn the per-kernel PMC listings of tools/gpu_r5_final.sh (tools/rocpd_pmc.py --by-grid over separate FETCH_SIZE / WRITE_SIZE
rocprofv3 passes of bench.py) into profiles/rNN_pmc_traffic.json: HBM bytes per launch for every kernel group of the step.

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE is reported in KB and on gfx950 tallies the
128-byte requests of wide coalesced reads at 64 B -> doubled; WRITE_SIZE is taken as reported (it equals the output size of
the conv kernels exactly).
usage: python tools/pmc_traffic.py gpurun_out/r4pmc profiles/rNN_pmc_traffic.json
"""
import json
import re
import sys


def parse(path, counter):
    out, key = {}, None
    for line in open(path):
        m = re.match(r"== (\S+)\s+grid_x=(\d+)\s+dispatches=(\d+)\s+mean_us=([\d.]+)(?:\s+size_rank=(\d+))?", line)
        if m:
            name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", m.group(1))
            key = (name, int(m.group(2)), int(m.group(5) or -1))
            out[key] = {"dispatches": int(m.group(3)), "mean_us": float(m.group(4))}
            continue
        m = re.match(r"\s+" + counter + r"\s+\d+\s+per-dispatch\s+(\d+)", line)
        if m and key:
            out[key][counter] = int(m.group(1))
    return out


SHORT = [("igemm_conv_s1_v9_kernelILi(\\d)ELi(\\d)ELi(\\d)ELi(\\d)", "conv_s1_v9<{0},{1},{2},epi={3}>"),
         ("igemm_down2s_kernelILi(\\d)ELi(\\d)ELi(\\d)ELi(\\d)ELb(\\d)", "down2s<{0},{1},{2},ext={3},stats={4}>"),
         ("igemm_wgrad_s1_v5_kernel", "wgrad_s1_v5"), ("igemm_wgrad_s2_v2_kernelILi(\\d)", "wgrad_s2<ext={0}>"),
         ("igemm_conv_mt_kernelILi(\\d)ELb(\\d)", "conv_s1_mt<wn={0},pipe={1}>"), ("igemm_conv_s1_v5_kernel", "conv_s1_tile"),
         ("igemm_up2_kernelILi(\\d)", "up2<{0}>"), ("igemm_down2_kernelILi(\\d)", "down2_tile<ext={0}>"),
         ("in_lrelu_seg_bwd_(\\w+?)_kernelILi(\\d)ELb(\\d)", "in_lrelu_seg_bwd_{0}<K={1},prior={2}>"),
         ("wgrad_c1_kernelILi4ELi8ELb(\\d)", "wgrad_c1<fused={0}>"), ("conv_c1_fwd_kernelILb(\\d)", "conv_c1_fwd<stats={0}>"),
         ("(\\w+?)_kernel", "{0}")]


def short(name):
    for pat, fmt in SHORT:
        m = re.match(pat, name)
        if m:
            return fmt.format(*m.groups())
    return name[:48]



def so_sha256():
    """sha256 of the liblnn_hip.so these counters were collected with: bench.py quotes the file only for the same binary."""
    import hashlib
    import os
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lifelong-nnunet_amd", "csrc", "liblnn_hip.so")
    return hashlib.sha256(open(so, "rb").read()).hexdigest()


def main(src, dst):
    f = parse(src + "/pmc_fetch.txt", "FETCH_SIZE")
    w = parse(src + "/pmc_write.txt", "WRITE_SIZE")
    rows = []
    # steps in the profiled run = launches of the image cast kernel (one per forward)
    nsteps = float(sum(a["dispatches"] for key, a in f.items() if key[0].startswith("cast_kernel")) or 3)
    for key, a in f.items():
        if key not in w or "FETCH_SIZE" not in a or "WRITE_SIZE" not in w[key] or a["mean_us"] < 40:
            continue
        byts = a["FETCH_SIZE"] * 1024 * 2 + w[key]["WRITE_SIZE"] * 1024
        rows.append({"kernel": short(key[0]), "grid_x": key[1], "size_rank": key[2], "launches_per_step": a["dispatches"] / nsteps,
                     "mean_us": a["mean_us"], "fetch_size_kb": a["FETCH_SIZE"], "write_size_kb": w[key]["WRITE_SIZE"],
                     "hbm_bytes_per_launch_corrected": byts, "hbm_tb_per_s": round(byts / a["mean_us"] / 1e6, 2)})
    rows.sort(key=lambda r: -r["mean_us"] * r["launches_per_step"])
    vox = 2 * 160 * 192 * 160

    def pick(kern, grid):        # the heaviest launch of that kernel and grid (= conv_blocks_localization.4.0)
        r = max((r for r in rows if r["kernel"] == kern and r["grid_x"] == grid), key=lambda r: r["mean_us"])
        alg = vox * (64 + 32) * 2
        return dict(r, algorithmic_bytes=alg, ratio_to_algorithmic=round(r["hbm_bytes_per_launch_corrected"] / alg, 3))
    out = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py itself "
                   "(tools/gpu_r5_final.sh, C2 step, weight gradients on the main stream, mean over the %d steps of the run)" % nsteps + "; FETCH_SIZE x2 per "
                   "MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE as reported; per launch",
           "layer": "conv_blocks_localization.4.0 64->32 @160x192x160 N=2 (algorithmic bytes 1.887 GB for each of the three)",
           "kernels": {"fwd": pick("conv_s1_v9<4,1,2,epi=1>", 131072), "dgrad": pick("conv_s1_v9<2,2,2,epi=0>", 131072),
                       "wgrad": pick("wgrad_s1_v5", 131072)},
           "all_kernels_over_40us": rows}
    out["so_sha256"] = so_sha256()
    json.dump(out, open(dst, "w"), indent=1)
    for k, v in out["kernels"].items():
        print(k, v["kernel"], v["mean_us"], v["hbm_bytes_per_launch_corrected"] / 1e9, v["ratio_to_algorithmic"], v["hbm_tb_per_s"])
    tot = sum(r["hbm_bytes_per_launch_corrected"] * r["launches_per_step"] for r in rows)
    print("HBM GB per step (kernels over 40 us):", tot / 1e9)


if __name__ == "__main__":
    main(*sys.argv[1:3])
