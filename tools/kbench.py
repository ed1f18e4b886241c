#!/usr/bin/env python
"""Micro-benchmark of the MFMA kernel families on single layers (HIP-event timing through the C-ABI).
    python tools/kbench.py [--layers dec4.0,enc0.1,...] [--iters 5] [--which fwd,dgrad,wgrad]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lifelong_nnunet_amd import native as nat

LAYERS = {  # name: (N, C, K, D, H, W, stride)
    "dec4.0": (2, 64, 32, 160, 192, 160, 1), "enc0.1": (2, 32, 32, 160, 192, 160, 1),
    "dec3.0": (2, 128, 64, 80, 96, 80, 1), "enc1.1": (2, 64, 64, 80, 96, 80, 1),
    "dec2.0": (2, 256, 128, 40, 48, 40, 1), "enc3.1": (2, 256, 256, 20, 24, 20, 1),
    "enc2.1": (2, 128, 128, 40, 48, 40, 1), "dec2.0d": (2, 128, 256, 40, 48, 40, 1),     # 128 input channels (v9 <8,1,1> since round 3)
    "dec4.0half": (2, 64, 32, 80, 96, 80, 1), "enc0.1half": (2, 32, 32, 80, 96, 80, 1),   # same layers, 1/8 of the voxels (fit the MALL)
    "enc1.0s2": (2, 32, 64, 160, 192, 160, 2), "enc2.0s2": (2, 64, 128, 80, 96, 80, 2),
    "enc3.0s2": (2, 128, 256, 40, 48, 40, 2), "enc4.0s2": (2, 256, 320, 20, 24, 20, 2),
    "enc4.1": (2, 320, 320, 10, 12, 10, 1), "dec0.0": (2, 640, 320, 10, 12, 10, 1), "enc5.1": (2, 320, 320, 5, 6, 5, 1),
    "dec1.0": (2, 512, 256, 20, 24, 20, 1),
    "enc5.0s2": (2, 320, 320, 10, 12, 10, 2), "enc3.1h": (2, 256, 256, 10, 12, 10, 1),
}
UPS = {  # transposed conv k2s2: name: (N, C, K, D, H, W) (low-res extents)
    "up4": (2, 64, 32, 80, 96, 80), "up3": (2, 128, 64, 40, 48, 40), "up2": (2, 256, 128, 20, 24, 20),
    "up1": (2, 320, 256, 10, 12, 10), "up0": (2, 320, 320, 5, 6, 5),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", default="dec4.0,enc0.1,dec3.0,enc1.1,dec2.0,enc1.0s2")
    ap.add_argument("--which", default="fwd,dgrad,wgrad")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--down2", type=int, default=-1, help="lnn_debug_force_down2_kernel: 0 tile kernel, 1 z-streaming kernel")
    ap.add_argument("--gen", type=int, default=-1, help="lnn_debug_set_gen_mode: 0 never the generic flattened-voxel kernels, 1 always, -1 automatic")
    ap.add_argument("--force", type=int, default=-1, help="lnn_debug_force_conv_kernel for the timed stride-1 calls (5, 7, 8, 9, 10 = macro-tile)")
    ap.add_argument("--phases", action="store_true", help="print the v3 conv kernel's per-phase cycle split")
    ap.add_argument("--check", default="", help="comma list of forced kernels (e.g. 5,9): run fwd/dgrad with each and compare outputs")
    ap.add_argument("--wgrad-phases", action="store_true", help="print the stride-1 wgrad kernel's per-phase cycle split (LNN_WGRAD_DEBUG=4)")
    a = ap.parse_args()
    dev = "cuda:0"
    nat.lib().lnn_debug_force_down2_kernel(a.down2)
    nat.lib().lnn_debug_set_gen_mode(a.gen)
    nat.lib().lnn_debug_force_conv_kernel(a.force)
    ws = torch.zeros(1 << 24, device=dev)          # split-K scratch the engine hands to the small layers
    wdbg = None
    if a.wgrad_phases:
        wdbg = torch.zeros(6, dtype=torch.int64, device=dev)
        os.environ["LNN_WGRAD_DEBUG"] = "4"
        os.environ["LNN_WGRAD_PHASEBUF"] = hex(wdbg.data_ptr())
    for name in a.layers.split(","):
        if name in UPS:
            N, C, K, D, H, W = UPS[name]
            x = (torch.randn((N, D, H, W, C), device=dev) * 0.5).half()
            y = torch.empty((N, 2 * D, 2 * H, 2 * W, K), dtype=torch.float16, device=dev)
            wf = torch.randn(nat.query("lnn_packed_weight_elems", 8, K, C), device=dev).half()
            fn = lambda: nat.call("lnn_convT3d_k2s2_fwd_ws", x, C, wf, y, K, N, D, H, W, C, K, ws, ws.numel())
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / a.iters * 1e-3
            gb = (x.numel() + y.numel()) * 2 / 1e9
            line = f"{name:9s} {C:4d}->{K:<4d} convT @{D}x{H}x{W} : fwd {t*1e3:7.3f} ms  {gb/t:6.0f} GB/s algorithmic"
            if "dgrad" in a.which:
                wdg = torch.randn(nat.query("lnn_packed_weight_elems", 8, C, K), device=dev).half()
                dxx = torch.empty_like(x)
                fd = lambda: nat.call("lnn_convT3d_k2s2_dgrad_ws", y, K, wdg, dxx, C, N, D, H, W, C, K, 0, ws, ws.numel())
                fd(); torch.cuda.synchronize()
                e0.record()
                for _ in range(a.iters):
                    fd()
                e1.record(); torch.cuda.synchronize()
                t = e0.elapsed_time(e1) / a.iters * 1e-3
                line += f" | dgrad {t*1e3:7.3f} ms"
            if "wgrad" in a.which:
                panel = torch.zeros(nat.query("lnn_wgrad_panel_elems", 8, C, K), device=dev)
                fw = lambda: nat.call("lnn_convT3d_k2s2_wgrad", x, C, y, K, panel, N, D, H, W, C, K)
                fw(); torch.cuda.synchronize()
                e0.record()
                for _ in range(a.iters):
                    fw()
                e1.record(); torch.cuda.synchronize()
                t = e0.elapsed_time(e1) / a.iters * 1e-3
                line += f" | wgrad {t*1e3:7.3f} ms  {gb/t:6.0f} GB/s"
            print(line, flush=True)
            del x, y
            continue
        cat = name.endswith("cat")            # e.g. dec4.0cat: the input as two separate 32-channel tensors
        N, C, K, D, H, W, s = LAYERS[name[:-3] if cat else name]
        Do, Ho, Wo = [(x - 1) // s + 1 for x in (D, H, W)]
        x = (torch.randn((N, D, H, W, C), device=dev) * 0.5).half()
        dy = (torch.randn((N, Do, Ho, Wo, K), device=dev) * 0.5).half()
        y = torch.empty_like(dy)
        dx = torch.empty_like(x)
        w = torch.randn((K, C, 3, 3, 3), device=dev) * 0.05
        b = torch.zeros(K, device=dev)
        wf = torch.empty(nat.query("lnn_packed_weight_elems", 27, K, C), dtype=torch.float16, device=dev)
        wd = torch.empty(nat.query("lnn_packed_weight_elems", 27, C, K), dtype=torch.float16, device=dev)
        nat.call("lnn_pack_weights", w, wf, 27, K, C, C * 27, 27, 1)
        nat.call("lnn_pack_weights", w, wd, 27, C, K, 27, C * 27, 1)
        panel = torch.zeros(nat.query("lnn_wgrad_panel_elems", 27, K, C), device=dev)
        flops = 2.0 * N * Do * Ho * Wo * C * K * 27
        xa = xb = dxa = dxb = None
        if cat:
            xa = x[..., :C // 2].contiguous(); xb = x[..., C // 2:].contiguous()
            dxa = torch.empty_like(xa); dxb = torch.empty_like(xb)
        st_mean, st_rstd = torch.zeros(N * K, device=dev), torch.zeros(N * K, device=dev)
        st_ws = torch.zeros(nat.query("lnn_instnorm_ws_doubles", N, K), dtype=torch.float64, device=dev)
        fns = {"fwd": (lambda: nat.call("lnn_conv3d_fwd_cat", xa, xb, C // 2, C // 2, wf, b, y, K, N, D, H, W, C, K)) if cat else
                      (lambda: nat.call("lnn_conv3d_fwd_g", x, C, wf, b, y, K, N, D, H, W, C, K, 3, 3, 3, s, s, s, ws, ws.numel())) if a.gen == 1 else
                      (lambda: nat.call("lnn_conv3d_fwd", x, C, wf, b, y, K, N, D, H, W, C, K, s)),
               "dgrad": (lambda: nat.call("lnn_conv3d_dgrad_cat_ws", dy, K, wd, dxa, dxb, C // 2, C // 2, N, D, H, W, C, K, 0, ws, ws.numel())) if cat else
                        (lambda: nat.call("lnn_conv3d_dgrad_ws", dy, K, wd, dx, C, N, D, H, W, C, K, s, 0, ws, ws.numel())),
               # conv + InstanceNorm statistics as the engine calls it (split-K workspace passed; includes the statistics pass where
               # the kernel has no fused epilogue)
               "fwd_st": lambda: nat.call("lnn_conv3d_fwd_in_stats", xa if cat else x, xb if cat else None, C // 2 if cat else C, C // 2 if cat else 0,
                                          wf, b, y, N, D, H, W, C, K, s, 1e-5, st_mean, st_rstd, st_ws, ws, ws.numel()),
               "wgrad": lambda: nat.call("lnn_conv3d_wgrad", x, C, dy, K, panel, N, D, H, W, C, K, s)}
        if s == 1 and C == K:
            # the stage's second block: its data gradient with / without pass 1 of the first block's normalisation backward in the epilogue
            V = D * H * W
            u = (torch.randn((N, D, H, W, C), device=dev) * 0.5).half()
            mean, rstd = torch.zeros(N * C, device=dev), torch.ones(N * C, device=dev)
            gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
            dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
            wsd = torch.zeros(nat.query("lnn_instnorm_ws_doubles", N, C), dtype=torch.float64, device=dev)
            fns["dgrad_red"] = lambda: nat.call("lnn_conv3d_dgrad_in_bwd_sums", dy, K, wd, dx, C, N, D, H, W, C, K, u, mean, rstd, gamma, beta,
                                                0.01, dg, db, 1.0, wsd, ws, ws.numel())

            def two_calls():
                nat.call("lnn_conv3d_dgrad_ws", dy, K, wd, dx, C, N, D, H, W, C, K, s, 0, ws, ws.numel())
                nat.call("lnn_instnorm_lrelu_bwd_sums", u, dx, C, N, V, C, mean, rstd, gamma, beta, 0.01, dg, db, 1.0, wsd)
            fns["dgrad_sums"] = two_calls
            fns["in_apply"] = lambda: nat.call("lnn_instnorm_lrelu_bwd_apply", u, dx, C, N, V, C, mean, rstd, gamma, beta, 0.01, wsd)
            fns["in_bwd"] = lambda: nat.call("lnn_instnorm_lrelu_bwd", u, dx, C, N, V, C, mean, rstd, gamma, beta, 0.01, dg, db, None, 1.0, wsd)
        out = []
        for k in a.which.split(","):
            if k not in fns:
                continue
            fn = fns[k]
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / a.iters * 1e-3
            out.append(f"{k} {t*1e3:7.3f} ms {flops/t/1e12:6.0f} TF/s")
        print(f"{name:9s} {C:4d}->{K:<4d} @{D}x{H}x{W} s{s} : " + " | ".join(out), flush=True)
        if a.check and s == 1:
            res = {}
            for which in [int(t) for t in a.check.split(",")]:
                nat.lib().lnn_debug_force_conv_kernel(which)
                y.zero_(); dx.zero_()
                fns["fwd"](); fns["dgrad"](); torch.cuda.synchronize()
                res[which] = (y.float().clone(), dx.float().clone())
            nat.lib().lnn_debug_force_conv_kernel(-1)
            ks = list(res)
            for k in ks[1:]:
                dfy = float((res[k][0] - res[ks[0]][0]).abs().max()); dfx = float((res[k][1] - res[ks[0]][1]).abs().max())
                print(f"   check kernel {k} vs {ks[0]}: fwd max|diff| {dfy:.3e} (max|y| {float(res[ks[0]][0].abs().max()):.2f})  "
                      f"dgrad max|diff| {dfx:.3e} (max|dx| {float(res[ks[0]][1].abs().max()):.2f})", flush=True)
            del res
        if wdbg is not None and s == 1 and "wgrad" in a.which:
            wdbg.zero_(); fns["wgrad"](); torch.cuda.synchronize()
            d = wdbg.cpu().tolist(); nt = max(d[5], 1)
            print("   wgrad phases (cycles per wave-tile): " + ", ".join(f"{n} {d[i]/nt:7.0f}" for i, n in enumerate(["prep(v5)|issue(v2)", "mfma", "vmwait(v5)|barrier1", "barrier(v5)|store", "barrier2"])) +
                  f"  | total {sum(d[:5])/nt:7.0f}  (112 MFMAs = 3584 pipe cycles per wave-tile)", flush=True)
        if a.phases and s == 1:
            dbg = torch.zeros(6, dtype=torch.int64, device=dev)
            nat.lib().lnn_debug_set_phase_buffer(dbg.data_ptr())
            fns["fwd"](); torch.cuda.synchronize()
            nat.lib().lnn_debug_set_phase_buffer(None)
            d = dbg.cpu().tolist()
            steps = max(d[5], 1)
            names = ["issue", "mfma", "store", "barrier"]
            print("   phases (cycles per wave-step): " + ", ".join(f"{n} {d[i]/steps:7.0f}" for i, n in enumerate(names)) +
                  f"  | total {sum(d[:4])/steps:7.0f}", flush=True)
        del x, dy, y, dx


if __name__ == "__main__":
    main()
