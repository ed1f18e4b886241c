#!/usr/bin/env python
"""Copy the evidence files of a `tools/gpu_rN_final.sh` lease from gpurun_out/<tag>/ into profiles/ under their per-round names.
    python tools/install_profiles.py r6final2 r06"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAP = {"bench_c2.json": "bench_c2.json", "kernel_stats.txt": "bench_c2_kernel_stats.txt",
       "kernel_stats_serialized.txt": "bench_c2_kernel_stats_serialized.txt", "layer_table.txt": "layer_table.txt",
       "layer_table_prostate.txt": "layer_table_prostate.txt", "library_gemm_roof.txt": "library_gemm_roof.txt",
       "pmc_fetch.txt": "pmc_fetch_by_kernel.txt", "pmc_write.txt": "pmc_write_by_kernel.txt", "pmc_mfma.txt": "pmc_mfma_by_kernel.txt",
       "pmc_mfma_clock.json": "pmc_mfma_clock.json", "pmc_traffic.json": "pmc_traffic.json", "timeline.txt": "step_timeline.txt",
       "rocprofv3_kernel_stats.csv": "rocprofv3_kernel_stats.csv"}


def main(tag, rnd):
    src = os.path.join(ROOT, "gpurun_out", tag)
    for a, b in MAP.items():
        if os.path.exists(os.path.join(src, a)):
            shutil.copy(os.path.join(src, a), os.path.join(ROOT, "profiles", f"{rnd}_{b}"))
            print("installed", f"profiles/{rnd}_{b}")
        else:
            print("MISSING", a)
    log = os.path.join(src, "pytest_gpu.log")
    if os.path.exists(log):
        lines = open(log).read().splitlines()
        sha = open(os.path.join(src, "so_sha256.txt")).read().split()[0] if os.path.exists(os.path.join(src, "so_sha256.txt")) else "?"
        with open(os.path.join(ROOT, "profiles", f"{rnd}_pytest_gpu.txt"), "w") as f:
            f.write(f"# python -m pytest tests -m gpu -q on one MI355X box, liblnn_hip.so sha256 {sha}\n")
            f.write("\n".join(l for l in lines if l.strip() and not l.startswith("  ")) + "\n")
        print("installed", f"profiles/{rnd}_pytest_gpu.txt")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
