#!/usr/bin/env python
"""Micro-benchmark of the HBM-bound InstanceNorm + LeakyReLU kernels (HIP-event timing through the C-ABI).
    python tools/kbench_in.py [--iters 5]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lifelong_nnunet_amd import native as nat

CASES = [(2, 32, 160 * 192 * 160), (2, 64, 80 * 96 * 80), (2, 128, 40 * 48 * 40), (2, 320, 10 * 12 * 10)]


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev = "cuda:0"
    for N, C, V in CASES:
        y = torch.randn((N, V, C), device=dev).half()
        z = torch.empty_like(y)
        dz = torch.randn((N, V, C), device=dev).half()
        mean = torch.zeros(N * C, device=dev); rstd = torch.ones(N * C, device=dev)
        g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
        dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev); dbias = torch.zeros(C, device=dev)
        ws = torch.zeros(nat.query("lnn_instnorm_ws_doubles", N, C), dtype=torch.float64, device=dev)
        nb = y.numel() * 2 / 1e9
        t1 = timeit(lambda: nat.call("lnn_instnorm_stats", y, N, V, C, 1e-5, mean, rstd, ws), a.iters)
        t2 = timeit(lambda: nat.call("lnn_instnorm_lrelu_fwd", y, z, C, N, V, C, mean, rstd, g, b, 0.01), a.iters)
        t3 = timeit(lambda: nat.call("lnn_instnorm_lrelu_bwd", y, dz, C, N, V, C, mean, rstd, g, b, 0.01, dg, db, dbias, 1.0, ws), a.iters)
        print(f"N={N} C={C:3d} V={V:8d}: stats {t1*1e3:6.3f} ms {nb/t1/1e3:5.2f} TB/s | fwd {t2*1e3:6.3f} ms {2*nb/t2/1e3:5.2f} TB/s | "
              f"bwd {t3*1e3:6.3f} ms {5*nb/t3/1e3:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
