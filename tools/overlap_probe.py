#!/usr/bin/env python
"""Can an HBM-bound normalisation pass run UNDER an MFMA-bound convolution if the convolution leaves CUs free?  (DESIGN.md section 9,
plan item 1.)  The z-streaming stride-1 conv (v9, one persistent 8-wave block per CU: it owns the CU's registers, nothing co-schedules)
is launched on stream A for ONE sample of enc0.1 (32 -> 32 channels at 160x192x160) with its grid sized for `budget` CUs; on stream B the
fused InstanceNorm + LeakyReLU forward of the OTHER sample's same-sized tensor.  Reported: each kernel alone, both back to back on one
stream, both concurrently on two streams -- HIP events on both streams, mean of `iters` repetitions.
    python tools/overlap_probe.py > profiles/r04_overlap_probe.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lifelong_nnunet_amd import native as nat

dev = "cuda:0"
N, C, K, D, H, W = 1, 32, 32, 160, 192, 160
V = D * H * W
x = (torch.randn((N, D, H, W, C), device=dev) * 0.5).half()
y = torch.empty((N, D, H, W, K), dtype=torch.float16, device=dev)
w = torch.randn((K, C, 3, 3, 3), device=dev) * 0.05
wf = torch.empty(nat.query("lnn_packed_weight_elems", 27, K, C), dtype=torch.float16, device=dev)
nat.call("lnn_pack_weights", w, wf, 27, K, C, C * 27, 27, 1)
b = torch.zeros(K, device=dev)
y2 = (torch.randn((N, D, H, W, K), device=dev)).half()
z2 = torch.empty_like(y2)
mean, rstd = torch.zeros(N * K, device=dev), torch.ones(N * K, device=dev)
gamma, beta = torch.ones(K, device=dev), torch.zeros(K, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def conv():
    nat.call("lnn_conv3d_fwd", x, C, wf, b, y, K, N, D, H, W, C, K, 1)


def norm():
    nat.call("lnn_instnorm_lrelu_fwd", y2, z2, K, N, V, K, mean, rstd, gamma, beta, 0.01)


def timed(fn_a, fn_b, iters=10):
    """fn_a on stream A, fn_b on stream B (either may be None), started together; returns ms until BOTH are done."""
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True)
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(iters):
        torch.cuda.synchronize()
        start.record()
        sa.wait_event(start); sb.wait_event(start)
        with torch.cuda.stream(sa):
            if fn_a: fn_a()
            ea.record(sa)
        with torch.cuda.stream(sb):
            if fn_b: fn_b()
            eb.record(sb)
        torch.cuda.synchronize()
        tot += max(start.elapsed_time(ea), start.elapsed_time(eb))
    return tot / iters


conv(); norm(); torch.cuda.synchronize()
t_norm = timed(None, norm)
print(f"# enc0.1-sized conv (1 sample, 543.6/2 GFLOP) on stream A, InstanceNorm+LeakyReLU forward of an equal tensor (630 MB moved) on stream B")
print(f"norm alone: {t_norm * 1e3:7.1f} us")
for budget in (0, 248, 240, 224, 208, 192):
    nat.lib().lnn_set_cu_budget(budget)
    conv(); torch.cuda.synchronize()
    t_conv = timed(conv, None)
    t_serial = timed(lambda: (conv(), norm()), None)
    t_both = timed(conv, norm)
    print(f"conv grid for {budget or 'all':>3} CUs: conv alone {t_conv * 1e3:7.1f} us | serial conv+norm {t_serial * 1e3:7.1f} us | "
          f"concurrent {t_both * 1e3:7.1f} us | hidden {100 * (t_serial - t_both) / t_norm:5.1f} % of the norm pass", flush=True)
nat.lib().lnn_set_cu_budget(0)
