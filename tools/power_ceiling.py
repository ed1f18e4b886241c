#!/usr/bin/env python
"""Is the ~0.43-of-peak plateau of the heaviest MFMA kernels a kernel limit or the chip's power limit?  (task 2.)

Runs three MFMA-bound kernels -- the z-streaming convolution (igemm_conv_s1_v9, dec4.0 64 -> 32 @ 160x192x160 forward), the stride-1
weight gradient (igemm_wgrad_s1_v5, same layer) and the vendor's fp16 GEMM (torch.matmul 8192^3 -> hipBLASLt) -- back to back on
ZERO-filled and on RANDOM operands: identical instruction streams, identical memory traffic, different switching activity in the
matrix pipes.  For each arm: wall time per launch (HIP events over >= 1 s of launches, so that DVFS settles), TFLOP/s, and the
board power / shader clock sampled from the driver's sysfs files (or rocm-smi) while the arm runs.

    python tools/power_ceiling.py [--seconds 1.5] [--fill zero|random|both] [--only conv,wgrad,gemm]

Under `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES` with one --fill per process the
same command gives the effective clock (GRBM_GUI_ACTIVE / duration) and the matrix-pipe busy fraction per arm."""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lifelong_nnunet_amd import native as nat


class Sampler:
    """Board power (W) and shader clock (MHz) every ~20 ms from sysfs; falls back to `rocm-smi --json` (slower) if unreadable."""

    def __init__(self):
        self.power_files = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")) + \
            sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
        self.clk_files = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
        self.sclk_files = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        self.samples = []
        self._stop = threading.Event()
        self._t = None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return f.read()
        except OSError:
            return None

    def _one(self):
        pw = [float(v) / 1e6 for v in (self._read(p) for p in self.power_files) if v and v.strip().isdigit()]
        ck = [float(v) / 1e6 for v in (self._read(p) for p in self.clk_files) if v and v.strip().isdigit()]
        if not ck:
            for p in self.sclk_files:
                txt = self._read(p) or ""
                for line in txt.splitlines():
                    if line.strip().endswith("*"):
                        try:
                            ck.append(float(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", "")))
                        except (IndexError, ValueError):
                            pass
        if pw or ck:
            return (max(pw) if pw else None, max(ck) if ck else None)
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out)
            card = next(iter(d.values()))
            p = next((float(v) for k, v in card.items() if "ower" in k and "(W)" in k), None)
            c = next((float(str(v).strip("()").lower().replace("mhz", "")) for k, v in card.items() if "sclk" in k.lower() and "clock speed" in k.lower()), None)
            return (p, c)
        except Exception:
            return (None, None)

    def _loop(self):
        while not self._stop.is_set():
            self.samples.append(self._one())
            time.sleep(0.02)

    def __enter__(self):
        self.samples = []
        self._stop.clear()
        self._t = threading.Thread(target=self._loop, daemon=True)
        self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._t.join()

    def summary(self):
        pw = [p for p, _ in self.samples if p is not None]
        ck = [c for _, c in self.samples if c is not None]
        # drop the first quarter (ramp)
        pw, ck = pw[len(pw) // 4:], ck[len(ck) // 4:]
        return {"power_w": round(sum(pw) / len(pw), 1) if pw else None, "sclk_mhz": round(sum(ck) / len(ck)) if ck else None,
                "samples": len(self.samples)}


def timed(fn, seconds):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    n = max(3, int(seconds * 1e3 / max(e0.elapsed_time(e1), 1e-3)))
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=1.5)
    ap.add_argument("--fill", default="both")
    ap.add_argument("--only", default="conv,wgrad,gemm")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    dev = "cuda:0"
    fills = ["zero", "random"] if a.fill == "both" else [a.fill]
    N, C, K, D, H, W = 2, 64, 32, 160, 192, 160
    flops_conv = 2.0 * N * D * H * W * C * K * 27
    sampler = Sampler()
    rows = []
    for fill in fills:
        rnd = fill == "random"
        mk = (lambda shape, s: (torch.randn(shape, device=dev) * s).half()) if rnd else (lambda shape, s: torch.zeros(shape, device=dev, dtype=torch.float16))
        arms = {}
        if "conv" in a.only or "wgrad" in a.only:
            x = mk((N, D, H, W, C), 0.5); dy = mk((N, D, H, W, K), 0.5); y = torch.empty_like(dy)
            w = torch.randn((K, C, 3, 3, 3), device=dev) * 0.05 if rnd else torch.zeros((K, C, 3, 3, 3), device=dev)
            b = torch.zeros(K, device=dev)
            wf = torch.empty(nat.query("lnn_packed_weight_elems", 27, K, C), dtype=torch.float16, device=dev)
            nat.call("lnn_pack_weights", w, wf, 27, K, C, C * 27, 27, 1)
            panel = torch.zeros(nat.query("lnn_wgrad_panel_elems", 27, K, C), device=dev)
            if "conv" in a.only:
                arms["conv_v9_fwd dec4.0"] = (lambda: nat.call("lnn_conv3d_fwd", x, C, wf, b, y, K, N, D, H, W, C, K, 1), flops_conv)
            if "wgrad" in a.only:
                arms["wgrad_v5 dec4.0"] = (lambda: nat.call("lnn_conv3d_wgrad", x, C, dy, K, panel, N, D, H, W, C, K, 1), flops_conv)
        if "gemm" in a.only:
            n = 8192
            ga = torch.randn((n, n), device=dev, dtype=torch.float16) if rnd else torch.zeros((n, n), device=dev, dtype=torch.float16)
            gb = torch.randn((n, n), device=dev, dtype=torch.float16) if rnd else torch.zeros((n, n), device=dev, dtype=torch.float16)
            gc = torch.empty((n, n), device=dev, dtype=torch.float16)
            arms["hipBLASLt gemm 8192^3"] = (lambda: torch.matmul(ga, gb, out=gc), 2.0 * n ** 3)
        for name, (fn, fl) in arms.items():
            with sampler:
                t, iters = timed(fn, a.seconds)
            s = sampler.summary()
            row = {"kernel": name, "fill": fill, "ms": round(t * 1e3, 4), "tflops": round(fl / t / 1e12, 1), "frac_of_2p5pf": round(fl / t / 2.5e15, 3),
                   "launches": iters, **s}
            rows.append(row)
            print(f"{name:24s} {fill:6s} {t*1e3:8.3f} ms {fl/t/1e12:7.1f} TFLOP/s ({fl/t/2.5e15:.3f} of peak)  power {s['power_w']} W  sclk {s['sclk_mhz']} MHz"
                  f"  ({iters} launches, {s['samples']} samples)", flush=True)
    by = {}
    for r in rows:
        by.setdefault(r["kernel"], {})[r["fill"]] = r
    for k, d in by.items():
        if "zero" in d and "random" in d:
            print(f"{k:24s} zero / random TFLOP/s = {d['zero']['tflops'] / d['random']['tflops']:.3f}", flush=True)
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    sys.exit(main())
