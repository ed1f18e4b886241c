#!/usr/bin/env python
"""One number per call: GB/s of lnn_gradnorm_sumsq (P = 31.2 M) and lnn_dice_ce_fwd (2 x 3 x 160x192x160) in THIS process
(the variants are read from the environment at first use: LNN_GRADNORM_VARIANT, LNN_DCE_BLOCKS)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lifelong_nnunet_amd import native as nat
import bench

dev = "cuda:0"
P = 31195600
g = torch.randn(P, device=dev) * 1e-3
ws = torch.zeros(nat.query("lnn_flat_reduce_ws_doubles"), dtype=torch.float64, device=dev)
t = bench.time_kernel(lambda: nat.call("lnn_gradnorm_sumsq", g, P, 1.0, ws, 1), iters=30)
ref = float((g.double() ** 2).sum())
B, K, V = 2, 3, 160 * 192 * 160
lg = torch.randn((B, K, V), device=dev)
lab = torch.randint(0, K, (B, V), device=dev).float()
out = torch.zeros(1, device=dev)
dws = torch.zeros(nat.query("lnn_dice_ce_ws_doubles", B, K), dtype=torch.float64, device=dev)
t2 = bench.time_kernel(lambda: nat.call("lnn_dice_ce_fwd", lg, lab, B, K, V, 0, 1e-5, out, dws), iters=30)
print(f"gradnorm variant {os.environ.get('LNN_GRADNORM_VARIANT', 'default')}: {t * 1e6:6.1f} us {4 * P / t / 1e9:6.0f} GB/s  (sumsq rel err "
      f"{abs(float(ws[0]) - ref) / ref:.1e}) | dice_ce_fwd blocks {os.environ.get('LNN_DCE_BLOCKS', 'default')}: {t2 * 1e6:6.1f} us "
      f"{(4 * K + 4) * B * V / t2 / 1e9:6.0f} GB/s loss {float(out):.6f}")
