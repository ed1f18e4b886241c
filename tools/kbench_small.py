#!/usr/bin/env python
"""Isolated timing of one ConvDropoutNormNonlin block of the two lowest levels, forward and backward, as ONE call per direction
(lnn_conv3d_fwd_in_lrelu / lnn_conv3d_dgrad_in_bwd: conv [-> split-K finalize] -> ONE normalisation launch) against the multi-launch
sequence (conv [-> split-K finalize] -> statistics -> finalize -> normalise / data gradient [-> finalize] -> reduce -> sums -> apply).  The multi-launch arm needs LNN_IN_SMALL=0 in the environment (the library reads the switch once), so run it twice:
    LNN_IN_SMALL=0 python tools/kbench_small.py ; python tools/kbench_small.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lifelong_nnunet_amd import native as nat  # noqa: E402

DEV = "cuda:0"
SHAPES = [("enc4.1 320->320 @10x12x10", 320, 320, 10, 12, 10, 1), ("enc4.0 256->320 s2 @20x24x20", 256, 320, 20, 24, 20, 2),
          ("enc5.1 320->320 @5x6x5", 320, 320, 5, 6, 5, 1), ("enc5.0 320->320 s2 @10x12x10", 320, 320, 10, 12, 10, 2)]


def timed(fn, iters=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    multi = os.environ.get("LNN_IN_SMALL") == "0"
    print(f"# {'multi-launch passes (LNN_IN_SMALL=0)' if multi else 'one normalisation launch per direction (default)'}; us per call")
    N = 2
    for name, C, K, D, H, W, s in SHAPES:
        torch.manual_seed(0)
        x = (torch.randn(N, D, H, W, C, device=DEV) * 0.5).half()
        w = torch.randn(K, C, 3, 3, 3, device=DEV) * (2.0 / (27 * C)) ** 0.5
        wp = torch.empty(nat.query("lnn_packed_weight_elems", 27, K, C), dtype=torch.float16, device=DEV)
        nat.call("lnn_pack_weights", w, wp, 27, K, C, C * 27, 27, 1)
        b, ga, be = torch.randn(K, device=DEV) * 0.1, torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
        Do, Ho, Wo = (D - 1) // s + 1, (H - 1) // s + 1, (W - 1) // s + 1
        V = Do * Ho * Wo
        y = torch.empty(N, Do, Ho, Wo, K, dtype=torch.float16, device=DEV); z = torch.empty_like(y)
        m, r = torch.empty(N * K, device=DEV), torch.empty(N * K, device=DEV)
        ws = torch.zeros(nat.query("lnn_instnorm_ws_doubles", N, max(C, K)), dtype=torch.float64, device=DEV)
        sk = torch.empty(64 * N * max(V, D * H * W) * 320, device=DEV)

        def fwd():
            if multi:
                nat.call("lnn_conv3d_fwd_in_stats", x, None, C, 0, wp, b, y, N, D, H, W, C, K, s, 1e-5, m, r, ws, sk, sk.numel())
                nat.call("lnn_instnorm_lrelu_fwd", y, z, K, N, V, K, m, r, ga, be, 0.01)
            else:
                nat.call("lnn_conv3d_fwd_in_lrelu", x, None, C, 0, wp, b, y, N, D, H, W, C, K, s, 1e-5, m, r, ga, be, 0.01, z, K, ws, sk, sk.numel())
        line = f"{name:32s} fwd {timed(fwd):7.1f}"
        if s == 1:
            wd = torch.empty(nat.query("lnn_packed_weight_elems", 27, C, K), dtype=torch.float16, device=DEV)
            nat.call("lnn_pack_weights", w, wd, 27, C, K, 27, C * 27, 1)
            dy = (torch.randn(N, D, H, W, K, device=DEV) * 0.5).half()
            u0 = (torch.randn(N, D, H, W, C, device=DEV)).half(); u = u0.clone()
            dx = torch.empty(N, D, H, W, C, dtype=torch.float16, device=DEV)
            mc, rc = torch.zeros(N * C, device=DEV), torch.ones(N * C, device=DEV)
            gc, bc, dg, db = torch.ones(C, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)

            def bwd():
                if multi:
                    nat.call("lnn_conv3d_dgrad_ws", dy, K, wd, dx, C, N, D, H, W, C, K, 1, 0, sk, sk.numel())
                    nat.call("lnn_instnorm_lrelu_bwd", u, dx, C, N, V, C, mc, rc, gc, bc, 0.01, dg, db, None, 1.0, ws)
                else:
                    nat.call("lnn_conv3d_dgrad_in_bwd", dy, K, wd, dx, C, N, D, H, W, C, K, u, mc, rc, gc, bc, 0.01, dg, db, 1.0, ws, sk, sk.numel())
            line += f"   dgrad + norm bwd {timed(bwd):7.1f}"
        print(line)


if __name__ == "__main__":
    main()
