#!/usr/bin/env python
"""How the fused normalisation-backward reduce (igemm_conv_v9.hip, EPI = 2: sum g u of the UN-normalised u in fp32 partials,
sum g xhat = rstd (sum g u - mean sum g) in fp64 afterwards) degrades when a channel's |mean| is large against its standard deviation
(ADVICE r4): the same sums from the unfused pass (which accumulates g xhat directly) and from an fp64 reference, for offsets 0 ... 100 std.
    python tools/probes/fused_reduce_conditioning.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from lifelong_nnunet_amd import native as nat
from tests.gpu_utils import DEV, pack_conv_dgrad, q16, to_cl_h

N, C, D, H, W = 2, 32, 24, 16, 24
K = C
g = torch.Generator().manual_seed(11)
base = torch.randn((N, C, D, H, W), generator=g)
gamma = 1 + 0.3 * torch.randn(C, generator=g)
beta = 0.2 * torch.randn(C, generator=g)
w = torch.randn((K, C, 3, 3, 3), generator=g) * 0.1
dy = torch.randn((N, K, D, H, W), generator=g)
V = D * H * W
print("offset/std   max rel err of sum g*xhat: fused vs fp64 | unfused vs fp64 | fused vs unfused     (relative to max |sum| over channels)")
for off in (0.0, 3.0, 10.0, 30.0, 100.0):
    u = q16(base + off)
    ub, _ = to_cl_h(u)
    dyb, _ = to_cl_h(dy)
    wp = pack_conv_dgrad(w.to(DEV))
    mean = torch.empty(N * C, device=DEV); rstd = torch.empty(N * C, device=DEV)
    nws = nat.query("lnn_instnorm_ws_doubles", N, C)
    ws0 = torch.zeros(nws, dtype=torch.float64, device=DEV)
    nat.call("lnn_instnorm_stats", ub, N, V, C, 1e-5, mean, rstd, ws0)
    ga, be = gamma.to(DEV), beta.to(DEV)
    res = []
    nat.lib().lnn_debug_force_conv_kernel(9)
    for fused in (False, True):
        dx = torch.zeros((N, D, H, W, C), dtype=torch.float16, device=DEV)
        ws = torch.zeros(nws, dtype=torch.float64, device=DEV)
        dg = torch.zeros(C, device=DEV); db = torch.zeros(C, device=DEV)
        if fused:
            nat.call("lnn_conv3d_dgrad_in_bwd_sums", dyb, K, wp, dx, C, N, D, H, W, C, K, ub, mean, rstd, ga, be, 0.01, dg, db, 1.0, ws, None, 0)
            assert nat.lib().lnn_debug_last_dgrad_reduce_fused() == 1
        else:
            nat.call("lnn_conv3d_dgrad_ws", dyb, K, wp, dx, C, N, D, H, W, C, K, 1, 0, None, 0)
            nat.call("lnn_instnorm_lrelu_bwd_sums", ub, dx, C, N, V, C, mean, rstd, ga, be, 0.01, dg, db, 1.0, ws)
        res.append((dx.clone(), ws[:N * C * 3].view(N * C, 3)[:, :2].cpu().clone()))
    nat.lib().lnn_debug_force_conv_kernel(-1)
    (dx0, s0), (dx1, s1) = res
    # fp64 reference of the sums from the kernel's own dz (fp16) and u
    dz = dx0.double().cpu().permute(0, 4, 1, 2, 3)
    ud = u.double()
    m = mean.double().cpu().view(N, C, 1, 1, 1); r = rstd.double().cpu().view(N, C, 1, 1, 1)
    xh = (ud - m) * r
    pre = gamma.double().view(1, C, 1, 1, 1) * xh + beta.double().view(1, C, 1, 1, 1)
    gg = dz * torch.where(pre > 0, torch.ones_like(pre), torch.full_like(pre, 0.01))
    ref = (gg * xh).sum((2, 3, 4)).reshape(-1)
    sc = float(ref.abs().max())
    print(f"{off:8.1f}     {float((s1[:, 1] - ref).abs().max()) / sc:12.3e}            {float((s0[:, 1] - ref).abs().max()) / sc:12.3e}        {float((s1[:, 1] - s0[:, 1]).abs().max()) / sc:12.3e}")
