#!/usr/bin/env python
"""Thread sweep of the CPU baseline (bench.py `cpu_baseline`: the oracle's run_iteration = forward, Dice+CE, backward, clip, SGD of
the 5-level U-Net in PyTorch CPU fp32): patches/s at 16 / 32 / 64 / 128 / all threads on a 128x128x128 sub-patch (two timed
iterations each after one warm-up), to show which thread count is the best this host does -- bench.py uses 32.
    python tools/cpu_thread_sweep.py > profiles/rNN_cpu_thread_sweep.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                            # noqa: E402
from oracle import losses as olosses, train as otrain                   # noqa: E402
from oracle.unet import OracleGenericUNet                               # noqa: E402
from lifelong_nnunet_amd.synthetic import make_patch_batch               # noqa: E402

shape, npool = (128, 128, 128), 5
data, tgts = make_patch_batch(1, shape, npool, seed=3)
w = olosses.ds_loss_weights(npool)
cores = os.cpu_count() or 1
cpu = "unknown"
try:
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            cpu = line.split(":", 1)[1].strip()
            break
except OSError:
    pass
print(f"# host: {cpu}, {cores} hardware threads; oracle.train.run_iteration on ONE {'x'.join(map(str, shape))} patch (B = 1), fp32")
print("threads   s/iteration   patches/s (128^3)   equivalent 160x192x160 patches/s")
ratio = (160 * 192 * 160) / (shape[0] * shape[1] * shape[2])
for nt in [t for t in (16, 32, 64, 128, 256) if t <= cores] + ([cores] if cores not in (16, 32, 64, 128, 256) else []):
    torch.set_num_threads(nt)
    torch.manual_seed(0)
    net = OracleGenericUNet(1, 32, 3, npool)
    opt = otrain.make_optimizer(net)
    otrain.run_iteration(net, opt, data, tgts, w)
    t0 = time.time()
    for _ in range(2):
        otrain.run_iteration(net, opt, data, tgts, w)
    dt = (time.time() - t0) / 2
    print(f"{nt:7d} {dt:13.2f} {1 / dt:19.4f} {1 / dt / ratio:33.4f}", flush=True)
