#!/usr/bin/env python
"""Calibration of the practical fp16 MFMA roof of THIS box: a plain library GEMM (torch.matmul -> hipBLASLt / rocBLAS) at sizes
that fit the MALL-less regime, timed with HIP events.  Not part of the product; it gives the roofline discussion in DESIGN.md a
measured reference next to the 2.5 PFLOP/s datasheet figure (run it under tools/gpu_r5_final.sh, which also collects
GRBM_GUI_ACTIVE / SQ_VALU_MFMA_BUSY_CYCLES for its kernels)."""
import sys

import torch


def main():
    dev = "cuda:0"
    for n in (4096, 8192, 16384):
        a = torch.randn((n, n), device=dev, dtype=torch.float16)
        b = torch.randn((n, n), device=dev, dtype=torch.float16)
        for _ in range(3):
            torch.matmul(a, b)
        torch.cuda.synchronize()
        iters = 20 if n <= 8192 else 6
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            torch.matmul(a, b)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / iters * 1e-3
        print(f"fp16 GEMM {n}^3: {t*1e3:8.3f} ms  {2.0*n**3/t/1e12:7.1f} TFLOP/s  ({2.0*n**3/t/2.5e15:.2f} of 2.5 PFLOP/s)", flush=True)
        del a, b


if __name__ == "__main__":
    sys.exit(main())
