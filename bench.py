#!/usr/bin/env python
"""Benchmark of the hot path: 3-D U-Net training iterations (nnUNetTrainerSequential.run_iteration) on
synthetic 160x192x160 patches -- BASELINE.json configs[1] -- one process per GPU.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --workload c3|c4|c5        (the continual-learning configurations of BASELINE.json, see WORKLOADS)

A "step" is one complete optimisation step (H2D-free: the patches are resident in HBM): forward, deep-supervised
Dice+CE (+ the trainer's regulariser), scaled backward, gradient all-reduce (N > 1), clip 12, SGD-Nesterov, head re-sync,
loss fetch.  Rank 0 prints ONE JSON line (see DESIGN.md "Measurement"); `config.h2d_inclusive` is the same loop with the
batches arriving from pinned host memory (copy of batch i+1 overlapped with step i), `regulariser_kernels` the HBM
roofline of the flat-arena / logits kernels at the workload's sizes.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_MFMA_F16_TFLOPS = 2500.0     # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md, chip-level table)

_C2 = {"patch_size": (160, 192, 160), "batch_size": 2, "num_pool": 5, "base_num_features": 32,
       "num_classes": 3, "num_input_channels": 1, "synthetic_period": 2}
WORKLOADS = {       # BASELINE.json configs[...]: (plans, trainer extension, description)
    "c2": (_C2, "sequential", "BASELINE configs[1]: nnUNetTrainerSequential.run_iteration"),
    "c1": ({**_C2, "patch_size": (40, 56, 40), "num_pool": 3}, "sequential", "BASELINE configs[0] shapes on the GPU (plumbing size)"),
    "c3": (_C2, "ewc", "BASELINE configs[2]: nnUNetTrainerEWC.run_iteration on the SECOND task (Dice+CE + EWC penalty over "
                       "P parameters, forward and backward)"),
    "c4": ({**_C2, "patch_size": (160, 160, 160)}, "lwf", "BASELINE configs[3]: nnUNetTrainerLWF.run_iteration, phase 3, one "
           "old head (reference semantics: one extra eval forward per head on its own batch, KL against the stored teacher logits)"),
    "c5": (_C2, "rehearsal_ewc", "BASELINE configs[4]: nnUNetTrainerRehearsalEWC.run_iteration on the second task "
           "(mixed-task batches from the fused case list + EWC penalty)"),
    # the reference's second use case (README.md:73, Task005_Prostate): an ANISOTROPIC 3d_fullres plan -- two modalities, in-plane
    # [1,3,3] kernels and [1,2,2] poolings in the first stages (nnUNetTrainerMultiHead.py:348-369 builds the network from these
    # lists).  Shape of upstream's Task005 plan as recalled (patch 20x320x256, batch 2, 6 poolings); runs on the generic-geometry
    # kernels (csrc/igemm_gen.hip): a measured number for that path, not a tuned one.
    "prostate": ({"patch_size": (20, 320, 256), "batch_size": 2, "num_pool": 6, "base_num_features": 32, "num_classes": 3,
                  "num_input_channels": 2, "synthetic_period": 2,
                  "pool_op_kernel_sizes": [[1, 2, 2], [1, 2, 2], [2, 2, 2], [2, 2, 2], [1, 2, 2], [1, 2, 2]],
                  "conv_kernel_sizes": [[1, 3, 3], [1, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3]]},
                 "sequential", "Task005_Prostate-shaped anisotropic 3d_fullres plan (2 modalities, [1,3,3] kernels / [1,2,2] poolings "
                 "in the first stages): nnUNetTrainerSequential.run_iteration"),
}


class ResidentBatches:
    """The reference's data dict, with tensors already in HBM (inputs resident when the timed region starts)."""

    def __init__(self, make_gen, device, n=1):
        """``make_gen``: a generator, or a zero-argument callable that builds one on first use (the validation generators of the
        benchmark trainers are never drawn from: their full-size synthetic batches are not generated at all)."""
        self._make, self._device, self._n = make_gen, device, n
        self._items = None
        self.i = 0
        if not callable(make_gen):
            self._fill()

    def _fill(self):
        gen = self._make() if callable(self._make) else self._make
        self._items = []
        for _ in range(self._n):
            d = next(gen)
            self._items.append({"data": d["data"].to(self._device), "target": [t.to(self._device) for t in d["target"]], "keys": d["keys"]})

    @property
    def items(self):
        if self._items is None:
            self._fill()
        return self._items

    def __iter__(self):
        return self

    def __next__(self):
        self.i += 1
        return self.items[self.i % len(self.items)]


class PrefetchingBatches:
    """H2D-inclusive variant (the reference's run_iteration starts with to_cuda, MH.py:606-617): batches live in PINNED host
    memory; the copy of batch i+1 runs on a side stream while step i computes (two device buffers, HIP events both ways)."""

    def __init__(self, items, device):
        import torch
        self.torch = torch
        self.host = [{"data": d["data"].cpu().pin_memory(), "target": [t.cpu().pin_memory() for t in d["target"]], "keys": d["keys"]}
                     for d in items]
        self.dev = [{"data": torch.empty_like(items[0]["data"], device=device),
                     "target": [torch.empty_like(t, device=device) for t in items[0]["target"]]} for _ in range(2)]
        self.copy_stream = torch.cuda.Stream(device=device)
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]
        self.i = 0
        self._issue(0, None)

    def _issue(self, i, after):
        torch = self.torch
        h, d = self.host[i % len(self.host)], self.dev[i % 2]
        if after is not None:
            self.copy_stream.wait_event(after)          # the buffer's previous consumer (step i-2) has been enqueued before `after`
        with torch.cuda.stream(self.copy_stream):
            d["data"].copy_(h["data"], non_blocking=True)
            for a, b in zip(d["target"], h["target"]):
                a.copy_(b, non_blocking=True)
            self.ready[i % 2].record(self.copy_stream)

    def __iter__(self):
        return self

    def __next__(self):
        torch = self.torch
        i = self.i
        main = torch.cuda.current_stream()
        main.wait_event(self.ready[i % 2])
        done_prev = torch.cuda.Event()
        done_prev.record(main)                           # everything of step i-1 (the other buffer's reader) is before this
        self._issue(i + 1, done_prev)
        self.i += 1
        d = self.dev[i % 2]
        return {"data": d["data"], "target": d["target"], "keys": self.host[i % len(self.host)]["keys"]}


def time_kernel(fn, iters=5):
    import torch
    fn(); torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()                      # events on torch's current stream = the stream the C-ABI launches on
    for _ in range(iters):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / iters * 1e-3


def kernel_rooflines(trainer):
    """Live HIP-event timing of the three MFMA kernel families on their heaviest launch (decoder level 4, block 0:
    64 -> 32 channels at full resolution = 24 % of all conv FLOPs)."""
    from lifelong_nnunet_amd import native as nat
    from lifelong_nnunet_amd.engine import ConvBlock
    eng = list(trainer.network._engines.values())[0]
    blk = max((b for b in eng.order if isinstance(b, ConvBlock) and b.cin > 1), key=lambda b: b.z.V * b.cin * b.cout)
    N, (D, H, W), C, K = eng.N, blk.in_dims, blk.cin, blk.cout
    flops = 2.0 * N * blk.z.V * C * K * 27
    res = {}
    if blk.x2 is not None:      # the engine keeps the two halves of the top-level concatenation as separate tensors
        fwd = lambda: nat.call("lnn_conv3d_fwd_cat", blk.x, blk.x2, blk.x.ld, blk.x.C, eng._wp(blk.wp_fwd), eng.pview(blk.b),
                               blk.y, K, N, D, H, W, C, K)
        dgrad = lambda: nat.call("lnn_conv3d_dgrad_cat", blk.y, K, eng._wp(blk.wp_dgrad), blk.gx, blk.gx2, blk.gx.ld, blk.gx.C,
                                 N, D, H, W, C, K, 0)
        wgrad = lambda: nat.call("lnn_conv3d_wgrad_cat", blk.x, blk.x2, blk.x.ld, blk.x.C, blk.y, K, eng._pn(blk.panel),
                                 N, D, H, W, C, K)
    else:
        fwd = lambda: nat.call("lnn_conv3d_fwd", blk.x, blk.x.ld, eng._wp(blk.wp_fwd), eng.pview(blk.b), blk.y, K,
                               N, D, H, W, C, K, blk.stride)
        dgrad = lambda: nat.call("lnn_conv3d_dgrad", blk.y, K, eng._wp(blk.wp_dgrad), blk.gx, blk.gx.ld, N, D, H, W,
                                 C, K, blk.stride, 0)
        wgrad = lambda: nat.call("lnn_conv3d_wgrad", blk.x, blk.x.ld, blk.y, K, eng._pn(blk.panel), N, D, H, W, C, K,
                                 blk.stride)
    res["igemm_conv_fwd"] = (flops, time_kernel(fwd))
    res["igemm_conv_dgrad"] = (flops, time_kernel(dgrad))
    res["igemm_wgrad"] = (flops, time_kernel(wgrad))
    return {"layer": f"{blk.prefix} {C}->{K} @{D}x{H}x{W} N={N}",
            "kernels": {k: {"tflops": f / t / 1e12, "ms": t * 1e3, "gflop": f / 1e9} for k, (f, t) in res.items()}}


def library_gemm_reference(n=8192, iters=20):
    """Calibration, not product: the vendor library's plain fp16 GEMM (torch.matmul -> hipBLASLt) on this box in this run, as a
    measured reference for what fraction of the 2.5 PFLOP/s datasheet peak dense fp16 MFMA code sustains here (the part lowers its
    clock under dense MFMA: profiles/r03_library_gemm_roof.txt)."""
    import torch
    a = torch.randn((n, n), device="cuda", dtype=torch.float16)
    b = torch.randn((n, n), device="cuda", dtype=torch.float16)
    for _ in range(3):
        torch.matmul(a, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.matmul(a, b)
    e1.record(); torch.cuda.synchronize()
    tf = 2.0 * n ** 3 / (e0.elapsed_time(e1) / iters * 1e-3) / 1e12
    return {"what": "torch.matmul fp16 %d^3 (hipBLASLt), same box, same run" % n, "tflops": tf, "frac_of_peak": tf / PEAK_MFMA_F16_TFLOPS}


def so_sha256():
    import hashlib
    from lifelong_nnunet_amd import native as nat
    return hashlib.sha256(open(nat.LIB_PATH, "rb").read()).hexdigest()


def committed_pmc(suffix, so_sha):
    """(file name, content) of the newest profiles/rNN_<suffix> collected with the library that is loaded now, else None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)), reverse=True):
        try:
            j = json.load(open(path))
        except Exception:
            continue
        if j.get("so_sha256") == so_sha:
            return os.path.basename(path), j
    return None


def regulariser_rooflines(trainer, plans):
    """HIP-event timing of the HBM-bound regulariser / optimiser kernels at this workload's sizes (P = flat arena size;
    logits of the full-resolution level), algorithmic bytes as SURVEY.md 8(d) counts them, against the 8 TB/s HBM peak."""
    import torch
    from lifelong_nnunet_amd import native as nat
    arena = trainer.network.arena
    P, dev = arena.size, arena.theta.device
    th, g = arena.theta, arena.grad
    ts, f, prev, score = (torch.rand(P, device=dev) for _ in range(4))
    out, ws = torch.zeros(1, device=dev), torch.zeros(nat.query("lnn_flat_reduce_ws_doubles"), dtype=torch.float64, device=dev)
    gs = torch.ones(1, device=dev)
    mom = torch.zeros(P, device=dev)
    thc, gc = th.clone(), torch.randn(P, device=dev) * 1e-3
    B, K = plans["batch_size"], plans["num_classes"]
    V = 1
    for d in plans["patch_size"]:
        V *= d
    lg = torch.randn((B, K, V), device=dev)
    lt = torch.randn((B, K, V), device=dev)
    lab = torch.randint(0, K, (B, V), device=dev).float()
    kl_out, kl_ws = torch.zeros(1, device=dev), torch.zeros(nat.query("lnn_kl_logits_ws_doubles", B), dtype=torch.float64, device=dev)
    dws = torch.zeros(nat.query("lnn_dice_ce_ws_doubles", B, K), dtype=torch.float64, device=dev)
    dl = torch.empty_like(lg)
    cases = {
        "ewc_penalty_fwd": (12 * P, lambda: nat.call("lnn_ewc_penalty_fwd", thc, ts, f, P, 0.4, out, ws)),
        "ewc_penalty_bwd": (20 * P, lambda: nat.call("lnn_ewc_penalty_bwd", thc, ts, f, P, 0.4, 1.0, gs, gc)),
        "fisher_square": (8 * P, lambda: nat.call("lnn_fisher_square", gc, f, P, 1.0)),
        "fisher_ema": (12 * P, lambda: nat.call("lnn_fisher_ema", gc, f, P, 1.0, 0.9)),
        "rw_update": (32 * P, lambda: nat.call("lnn_rw_update", thc, prev, gc, f, score, P, 1.0, 12.0, ws, 0.9, 1e-8, 1)),
        "gradnorm_sumsq": (4 * P, lambda: nat.call("lnn_gradnorm_sumsq", gc, P, 1.0, ws, 1)),
        "sgd_nesterov_clipped": (20 * P, lambda: nat.call("lnn_sgd_nesterov_step_clipped", thc, mom, gc, P, 1e-3, 0.99, 3e-5, 1.0, 12.0, ws)),
        "kl_logits (LwF, full-res logits)": (2 * 4 * B * K * V, lambda: nat.call("lnn_kl_logits", lg, lt, B, K, V, 2.0, kl_out, kl_ws)),
        "dice_ce_fwd (full-res level)": ((4 * K + 4) * B * V, lambda: nat.call("lnn_dice_ce_fwd", lg, lab, B, K, V, 0, 1e-5, out, dws)),
        "dice_ce_bwd (full-res level)": ((8 * K + 4) * B * V, lambda: nat.call("lnn_dice_ce_bwd", lg, lab, B, K, V, 0, 1e-5, dws, 1.0, None, 1.0, dl)),
    }
    res = {}
    for name, (nbytes, fn) in cases.items():
        try:
            t = time_kernel(fn, iters=10)
            res[name] = {"ms": t * 1e3, "algorithmic_MB": nbytes / 1e6, "GBps": nbytes / t / 1e9, "frac_of_8TBps": nbytes / t / 8e12}
        except Exception as e:                     # an entry this build does not export must not hide the others
            res[name] = {"error": repr(e)}
    return {"P": P, "logits_elems": B * K * V, "kernels": res}


def cpu_baseline_and_parity(tr, plans, flops_full, batch, sample="full"):
    """The oracle (pure PyTorch CPU fp32 restatement of the reference's step, oracle/) on the GPU box's host cores, timed on
    ONE full-size patch (B=1; `sample="small"`: a 128^3 sub-patch scaled by the voxel ratio), with the weights the benchmarked
    trainer holds after its timed steps -- and the SAME patch through the HIP path: the Dice half of BASELINE.json's metric
    (`mean Dice vs ref`) and the loss gate the reference prints with every timing (MH.py:1017-1019).
    Returns (cpu_baseline, parity)."""
    import torch
    from oracle import losses as olosses, train as otrain
    from oracle.unet import OracleGenericUNet
    from lifelong_nnunet_amd.synthetic import make_patch_batch
    cores_present = os.cpu_count() or 1
    cores = min(cores_present, 32)           # oneDNN conv3d stops scaling (and thrashes) far below 256 threads
    torch.set_num_threads(cores)
    K, npool = plans["num_classes"], plans["num_pool"]
    data = batch["data"][:1].float().cpu()
    tgts = [t[:1].float().cpu() for t in batch["target"]]
    ratio = 1.0
    if sample != "full":
        q = 2 ** npool
        sub = tuple(max(2 * q, min(128, p // q * q)) for p in plans["patch_size"])
        data = data[:, :, :sub[0], :sub[1], :sub[2]].contiguous()
        tgts = [t[:, :, :sub[0] >> i, :sub[1] >> i, :sub[2] >> i].contiguous() for i, t in enumerate(tgts)]
        for a_, b_ in zip(sub, plans["patch_size"]):
            ratio *= a_ / b_
    shape = tuple(data.shape[2:])
    net = OracleGenericUNet(1, plans["base_num_features"], K, npool)
    net.load_state_dict({k: v.detach().float().cpu() for k, v in tr.network.state_dict().items()})
    w = olosses.ds_loss_weights(npool)
    # HIP path on the same patch (no-grad forward, B = 1 engine), BEFORE the oracle's optimiser step changes its weights
    from lifelong_nnunet_amd.losses import DC_and_CE_loss, MultipleOutputLoss2, ds_loss_weights
    loss_fn = MultipleOutputLoss2(DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False}, {}), ds_loss_weights(npool))
    dev = tr.network.device_
    with torch.no_grad():
        tr.network.eval()
        out_g = tr.network(data.to(dev))
        loss_g = float(loss_fn(out_g, [t.to(dev) for t in tgts]))
        seg_g = out_g[0].argmax(1).cpu()
        tr.network.train()
    del out_g
    opt = otrain.make_optimizer(net)
    small = make_patch_batch(1, tuple(2 ** (npool + 1) for _ in shape), npool, seed=2)
    warm = OracleGenericUNet(1, plans["base_num_features"], K, npool)
    otrain.run_iteration(warm, otrain.make_optimizer(warm), small[0], small[1], w)   # warm-up (thread pools, oneDNN primitives)
    del warm
    t0 = time.time()
    loss_o, out_o = otrain.run_iteration(net, opt, data, tgts, w)
    dt = time.time() - t0
    seg_o = out_o[0].detach().argmax(1)
    lab = tgts[0][:, 0].long()

    def dice(a_, b_):
        ds = []
        for c in range(1, K):
            x, y = a_ == c, b_ == c
            den = int(x.sum()) + int(y.sum())
            if den:
                ds.append(2.0 * int((x & y).sum()) / den)
        return sum(ds) / len(ds) if ds else float("nan")

    d_go, d_gl, d_ol = dice(seg_g, seg_o), dice(seg_g, lab), dice(seg_o, lab)
    parity = {"patch": "x".join(map(str, shape)) + ", B=1, the trainer's weights after the timed steps",
              "loss_hip": loss_g, "loss_oracle": loss_o, "loss_rel_err": abs(loss_g - loss_o) / max(abs(loss_o), 1e-12),
              "dice_hip_vs_oracle_seg": d_go, "mean_dice_hip": d_gl, "mean_dice_oracle": d_ol, "abs_delta_dice": abs(d_gl - d_ol),
              "voxel_agreement": float((seg_g == seg_o).float().mean()),
              "gates": {"loss_rel_err<=1e-4": abs(loss_g - loss_o) <= 1e-4 * abs(loss_o), "abs_delta_dice<=1e-3": abs(d_gl - d_ol) <= 1e-3,
                        "dice_hip_vs_oracle_seg>=0.9": d_go >= 0.9}}
    cpu_name = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_name = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    base = {"value": ratio / dt, "unit": "patches/s", "cores": cores, "cores_used": cores, "cores_present": cores_present,
            "kind": "port", "extrapolated": sample != "full", "cpu": cpu_name,
            "sample": f"oracle.train.run_iteration (forward, Dice+CE, backward, clip, SGD), same {npool}-level U-Net, ONE "
                      f"{'x'.join(map(str, shape))} patch (B=1) in {dt:.1f} s" + ("" if sample == "full" else f", scaled by the voxel ratio {ratio:.4f}"),
            "gflops": flops_full * ratio / dt / 1e9}
    return base, parity


def regulariser_parity(tr, ext):
    """Bench-size parity gate of the continual-learning term of c3 / c4 / c5 (relative loss error <= 1e-4
    beside the timing): the HIP value on the trainer's OWN state against the oracle's CPU restatement on the same numbers.
      ewc / rehearsal_ewc: penalty lambda/2 sum F (theta - theta*)^2 over all P parameters and the norm of its gradient
                           (deep_supervision.py:58-83) -- lnn_ewc_penalty_fwd / _bwd vs oracle.losses.ewc_penalty + autograd;
      lwf:                 batchmean KL at temperature T of one full-size logits pair (deep_supervision.py:194-196) --
                           lnn_kl_logits vs oracle.losses.lwf_distillation."""
    import torch
    from oracle import losses as ol
    if ext in ("ewc", "rehearsal_ewc"):
        named = list(tr.network.named_parameters())
        arena = tr.network.arena
        keep = arena.grad.clone()
        arena.grad.zero_()
        tr.loss.update_network_params(iter(named))
        zero = torch.zeros((), device=arena.theta.device, requires_grad=True)     # the penalty alone: base loss 0
        pen = tr.loss._regularised(zero, tr.loss.ewc_lambda)
        pen.backward()
        torch.cuda.synchronize()
        v_hip, g_hip = float(pen), float(arena.grad.double().norm())
        arena.grad.copy_(keep)
        tr.loss.update_network_params(tr.network.named_parameters())
        cpu = [(n, p.detach().cpu().clone().requires_grad_(True)) for n, p in named]
        fisher = {t: {n: v.detach().float().cpu() for n, v in d.items()} for t, d in tr.fisher.items()}
        stars = {t: {n: v.detach().float().cpu() for n, v in d.items()} for t, d in tr.params.items()}
        ref = ol.ewc_penalty(cpu, fisher, stars, tr.loss.ewc_lambda, first_task_only=True)
        ref.backward()
        v_ref = float(ref)
        g_ref = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for _, p in cpu)))
        rv, rg = abs(v_hip - v_ref) / max(abs(v_ref), 1e-30), abs(g_hip - g_ref) / max(abs(g_ref), 1e-30)
        return {"what": "EWC penalty and its gradient norm over P = %d parameters, HIP vs oracle.losses.ewc_penalty (CPU fp32)" % arena.size,
                "penalty_hip": v_hip, "penalty_oracle": v_ref, "penalty_rel_err": rv, "grad_norm_hip": g_hip,
                "grad_norm_oracle": g_ref, "grad_norm_rel_err": rg, "gates": {"rel_err<=1e-4": rv <= 1e-4 and rg <= 1e-4}}
    if ext == "lwf":
        from lifelong_nnunet_amd.losses import kl_logits
        pred, teach = tr.LwFloss.pred_logits[0], tr.LwFloss.target_logits[0]
        T = tr.LwFloss.lwf_temperature
        v_hip = float(kl_logits(pred, teach, T))
        v_ref = float(ol.lwf_distillation(pred.detach().float().cpu(), teach.detach().float().cpu(), T))
        rv = abs(v_hip - v_ref) / max(abs(v_ref), 1e-30)
        return {"what": "LwF distillation KL (T = %g) of the last iteration's full-size logits pair %s, lnn_kl_logits vs "
                        "oracle.losses.lwf_distillation (CPU fp32)" % (T, "x".join(map(str, pred.shape))),
                "kl_hip": v_hip, "kl_oracle": v_ref, "rel_err": rv, "gates": {"rel_err<=1e-4": rv <= 1e-4}}
    return None


def iteration_parity(tr, ext, plans):
    """Bench-size gate on the FULL loss of the continual-learning iteration (relative loss error <= 1e-4 beside
    the c3 / c4 / c5 timings): one full-size patch (B = 1) of the trainer's own resident batches through the fp16 MFMA engine with
    the trainer's weights after the timed steps and through the TRAINER'S loss object -- Dice+CE over the deep-supervision levels
    + the live EWC penalty (deep_supervision.py:58-83) or + the LwF distillation KL of the stored logits pair
    (deep_supervision.py:185-214) -- against the oracle's CPU fp32 forward of the same network on the same patch + the oracle's
    restatement of the same terms."""
    import torch
    from oracle import losses as ol, train as otrain
    from oracle.unet import OracleGenericUNet
    batch = tr.tr_gen.items[0]
    data = batch["data"][:1].float()
    tgts = [t[:1].float() for t in batch["target"]]
    K, npool = plans["num_classes"], plans["num_pool"]
    dev = tr.network.device_
    named = list(tr.network.named_parameters())
    from lifelong_nnunet_amd.losses import DC_and_CE_loss, MultipleOutputLoss2, ds_loss_weights
    plain = MultipleOutputLoss2(DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False}, {}), ds_loss_weights(npool))
    with torch.no_grad():
        tr.network.eval()
        out_g = tr.network(data.to(dev))
        tg_g = [t.to(dev) for t in tgts]
        base_g = float(plain(out_g, tg_g))                 # the Dice+CE term alone, same kernels the trainer's loss object runs
        if ext in ("ewc", "rehearsal_ewc"):
            tr.loss.update_network_params(iter(named))
            loss_g = float(tr.loss(out_g, tg_g))
            tr.loss.update_network_params(tr.network.named_parameters())
        else:
            loss_g = float(tr.LwFloss(out_g, tg_g))
        tr.network.train()
    del out_g, tg_g
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    net = OracleGenericUNet(1, plans["base_num_features"], K, npool)
    net.load_state_dict({k: v.detach().float().cpu() for k, v in tr.network.state_dict().items()})
    with torch.no_grad():
        out_o = net(data.cpu())
        tg_o = [t.cpu() for t in tgts]
        w = ol.ds_loss_weights(npool)
        base = ol.multiple_output_loss(out_o, tg_o, w)
        # size of the Dice+CE terms, sum_i w_i (CE_i + |Dice_i|): the term is a difference of O(1) quantities (see plain_loss_parity)
        terms = 0.0
        for i in range(len(out_o)):
            if w[i] != 0:
                dl = float(ol.soft_dice_loss(out_o[i], tg_o[i]))
                terms += float(w[i]) * (float(ol.dc_and_ce_loss(out_o[i], tg_o[i])) - 2.0 * dl)
        if ext in ("ewc", "rehearsal_ewc"):
            cpu = [(n, p.detach().cpu()) for n, p in named]
            fisher = {t: {n: v.detach().float().cpu() for n, v in d.items()} for t, d in tr.fisher.items()}
            stars = {t: {n: v.detach().float().cpu() for n, v in d.items()} for t, d in tr.params.items()}
            loss_o = float(base + ol.ewc_penalty(cpu, fisher, stars, tr.loss.ewc_lambda, first_task_only=True))
            what = "Dice+CE over the deep-supervision levels + the EWC penalty of the first previous task"
        else:
            preds = [p.detach().float().cpu() for p in tr.LwFloss.pred_logits]
            teach = [t.detach().float().cpu() for t in tr.LwFloss.target_logits]
            loss_o = float(otrain.lwf_loss_value(base, preds, teach, tr.LwFloss.lwf_temperature))
            what = "Dice+CE over the deep-supervision levels + the distillation KL of the last iteration's logits pair(s)"
    rel = abs(loss_g - loss_o) / max(abs(loss_o), 1e-30)
    # The two terms are gated SEPARATELY: the regulariser can be orders of magnitude larger than Dice+CE (the batchmean KL over 4.1 M
    # voxels is ~1.7e6 against a Dice+CE of O(1)), and a gate on the sum alone would pass with a Dice+CE term that is 100 % off.
    base_o = float(base)
    reg_g, reg_o = loss_g - base_g, loss_o - base_o
    dice_ce_err = abs(base_g - base_o)
    reg_rel = abs(reg_g - reg_o) / max(abs(reg_o), 1e-30)
    # the regulariser read off as (total - Dice+CE) carries the fp32 rounding of the total: not resolvable below eps * |total|
    reg_floor = 2.0 ** -22 * max(abs(loss_o), 1e-30) / max(abs(reg_o), 1e-30)
    return {"what": what + ": ONE %s patch (B = 1), the trainer's weights after the timed steps, fp16 MFMA engine + the trainer's "
                           "loss object vs the oracle's CPU fp32 forward + restated terms" % "x".join(map(str, data.shape[2:])),
            "loss_hip": loss_g, "loss_oracle": loss_o, "loss_rel_err": rel,
            "dice_ce_hip": base_g, "dice_ce_oracle": base_o, "dice_ce_rel_err": dice_ce_err / max(abs(base_o), 1e-30),
            "dice_ce_err_rel_to_terms": dice_ce_err / max(terms, 1e-30), "sum_of_dice_ce_term_magnitudes": terms,
            "regulariser_hip": reg_g, "regulariser_oracle": reg_o, "regulariser_rel_err": reg_rel,
            "gates": {"loss_rel_err<=1e-4": rel <= 1e-4, "dice_ce_err<=1e-4*(CE+|Dice|)": dice_ce_err <= 1e-4 * terms,
                      "regulariser_rel_err<=1e-4": reg_rel <= 1e-4 + reg_floor}}


def plain_loss_parity(tr, plans):
    """Deep-supervised Dice+CE of ONE full-size patch (B = 1): fp16 MFMA engine with the trainer's weights vs the oracle's CPU fp32
    forward of the same plan-built network (any plan: input channels, per-level poolings / kernel extents); gate 1e-4."""
    import torch
    from oracle import losses as ol
    from oracle.unet import OracleGenericUNet
    from lifelong_nnunet_amd.losses import DC_and_CE_loss, MultipleOutputLoss2, ds_loss_weights
    batch = tr.tr_gen.items[0]
    data = batch["data"][:1].float()
    tgts = [t[:1].float() for t in batch["target"]]
    npool = plans["num_pool"]
    dev = tr.network.device_
    loss_fn = MultipleOutputLoss2(DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False}, {}), ds_loss_weights(npool))
    with torch.no_grad():
        tr.network.eval()
        loss_g = float(loss_fn(tr.network(data.to(dev)), [t.to(dev) for t in tgts]))
        tr.network.train()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    net = OracleGenericUNet(plans["num_input_channels"], plans["base_num_features"], plans["num_classes"], npool,
                            pool_op_kernel_sizes=plans.get("pool_op_kernel_sizes"), conv_kernel_sizes=plans.get("conv_kernel_sizes"))
    net.load_state_dict({k: v.detach().float().cpu() for k, v in tr.network.state_dict().items()})
    with torch.no_grad():
        out_o = net(data.cpu())
        tg = [t.cpu() for t in tgts]
        w = ol.ds_loss_weights(npool)
        loss_o = float(ol.multiple_output_loss(out_o, tg, w))
        # the loss is a DIFFERENCE of O(1) terms (cross-entropy >= 0, soft Dice in [-1, 0]) and comes close to zero while training
        # (0.18 on this plan after a few steps): the error is gated against the size of the terms, sum_i w_i (CE_i + |Dice_i|), and
        # reported against |loss| as well
        terms = 0.0
        for i in range(len(out_o)):
            if w[i] != 0:
                dl = float(ol.soft_dice_loss(out_o[i], tg[i]))
                terms += float(w[i]) * (float(ol.dc_and_ce_loss(out_o[i], tg[i])) - 2.0 * dl)
    err = abs(loss_g - loss_o)
    rel = err / max(abs(loss_o), 1e-30)
    return {"what": "deep-supervised Dice+CE of ONE %s patch (B = 1, %d channels), the trainer's weights after the timed steps: fp16 "
                    "MFMA engine vs the oracle's CPU fp32 forward" % ("x".join(map(str, data.shape[2:])), data.shape[1]),
            "loss_hip": loss_g, "loss_oracle": loss_o, "loss_rel_err": rel, "sum_of_term_magnitudes": terms,
            "loss_err_rel_to_terms": err / max(terms, 1e-30),
            "gates": {"loss_err<=1e-4*(CE+|Dice|)": err <= 1e-4 * terms, "loss_rel_err<=1e-3": rel <= 1e-3}}


def build_trainer(workload, device, rank):
    """Trainer of one BASELINE.json configuration in the state its timed iteration needs (second task for the CL methods)."""
    import torch
    from lifelong_nnunet_amd import get_trainer_class
    plans, ext, wl_desc = WORKLOADS[workload]
    plans = dict(plans)
    Trainer = get_trainer_class(ext)

    def provider(task, split, p):
        from lifelong_nnunet_amd.training.network_training.multihead.nnUNetTrainerMultiHead import default_data_provider
        # TWO distinct resident batches, alternated: no step sees the patch of the step before it
        make = lambda: default_data_provider(task, split, p, seed=12345 + 7919 * rank)
        return ResidentBatches(make() if split == "train" else make, device, n=2)

    kw = {"cases_per_task": 8} if ext == "rehearsal_ewc" else {}
    tr = Trainer("seg_outputs", "synthetic_task_A", plans=plans, data_provider=provider, device=device, fold=0, **kw)
    tr.initialize(True, num_epochs=1000)
    tr.network.train()
    extra_cfg = {}
    if ext in ("ewc", "rehearsal_ewc"):
        # a finished first task: Fisher / theta* dictionaries over all P parameters (values are irrelevant for the timing)
        g = torch.Generator(device=device).manual_seed(1)
        named = list(tr.network.named_parameters())
        tr.fisher["synthetic_task_A"] = {n: torch.rand(p.shape, device=device, generator=g) for n, p in named}
        tr.params["synthetic_task_A"] = {n: p.detach().clone() + 1e-3 * torch.randn(p.shape, device=device, generator=g) for n, p in named}
        tr.mh_network.add_new_task("synthetic_task_B", use_init=True)
        tr.network = tr.mh_network.assemble_model("synthetic_task_B")
        tr.task = "synthetic_task_B"
        if ext == "rehearsal_ewc":
            gen, _ = tr.get_basic_generators()            # fused case list: task B + a seeded 25 % of task A
            tr.tr_gen = ResidentBatches(gen, device, n=4)
            extra_cfg["fused_train_cases"] = len(tr.dataset_tr)
            extra_cfg["batches_with_a_rehearsed_case"] = sum(any(str(k).startswith("synthetic_task_A") for k in it["keys"]) for it in tr.tr_gen.items)
        tr.loss.update_ewc_params(tr.fisher, tr.params)
        tr.loss.update_network_params(tr.network.named_parameters())
        extra_cfg["penalty_params"] = tr.network.arena.size
    if ext == "lwf":
        from lifelong_nnunet_amd.training.network_training.lwf.nnUNetTrainerLWF import calculate_target_logits
        tr.mh_network.add_new_task("synthetic_task_B", use_init=True)
        tr.network = tr.mh_network.assemble_model("synthetic_task_B", freeze_body=False)
        tr.task, tr.num_batches_per_epoch = "synthetic_task_B", 2
        tr.target_logits = calculate_target_logits(tr.mh_network, tr.tr_gen, tr.num_batches_per_epoch, True)   # teacher store stays in HBM
        tr.network.train()
        tr.freeze_run, tr.loss, tr.batch_idx = False, tr.LwFloss, 0
        extra_cfg["teacher_store_MB"] = sum(t.numel() * t.element_size() for v in tr.target_logits.values() for t in v) / 1e6
    return tr, plans, ext, wl_desc, extra_cfg


def heaviest_block(eng):
    from lifelong_nnunet_amd.engine import ConvBlock
    return max((b for b in eng.order if isinstance(b, ConvBlock) and b.cin > 1), key=lambda b: b.z.V * b.cin * b.cout)


def other_workload(workload, args, device, rank):
    """One of the continual-learning configurations as extra keys of the same line (N = 1): same loop, same clock."""
    import torch
    tr, plans, ext, wl_desc, extra_cfg = build_trainer(workload, device, rank)
    tr.defer_loss_fetch = True          # as the headline loop (see main)
    for _ in range(args.warmup):
        tr.run_iteration(tr.tr_gen, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = tr.run_iteration(tr.tr_gen, True)
    loss = float(loss)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    B = plans["batch_size"]
    res = {"workload": f"{wl_desc}, {'x'.join(map(str, plans['patch_size']))} patches, batch {B}", "value": B * args.steps / dt,
           "unit": "patches/s", "ms_per_step": dt / args.steps * 1e3, "steps": args.steps, "loss": float(loss)}
    res.update(extra_cfg)
    if not args.no_cpu_baseline and ext in ("ewc", "rehearsal_ewc", "lwf"):
        try:
            res["parity"] = regulariser_parity(tr, ext)
        except Exception as e:
            res["parity"] = {"error": repr(e)}
        try:
            res["parity"]["full_iteration"] = iteration_parity(tr, ext, plans)
        except Exception as e:
            res["parity"]["full_iteration"] = {"error": repr(e)}
    elif not args.no_cpu_baseline:
        try:
            res["parity"] = plain_loss_parity(tr, plans)
        except Exception as e:
            res["parity"] = {"error": repr(e)}
    eng = list(tr.network._engines.values())[0]
    fl, _ = eng.flops_per_patch()
    res["conv_gflop_per_patch"] = fl / 1e9
    res["conv_stack_frac_of_mfma_peak"] = res["value"] * fl / 1e12 / PEAK_MFMA_F16_TFLOPS
    if ext == "lwf":          # the fix behind a flag: every head evaluated on the training batch (one batch, one body pass)
        tr.same_batch_predictions = True
        for _ in range(2):
            tr.run_iteration(tr.tr_gen, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tr.run_iteration(tr.tr_gen, True)
        torch.cuda.synchronize()
        res["same_batch_predictions_patches_per_s"] = B * args.steps / (time.perf_counter() - t0)
        res["note"] = "value = reference semantics (T+2 batches per iteration, one extra eval forward per head on its own batch)"
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the warm-up and the timed steps (no synchronising-fetch / H2D-inclusive repeats): counter passes")
    ap.add_argument("--cpu-sample", default="full", choices=["full", "small"],
                    help="CPU baseline / parity patch: one full-size patch (~15 s of CPU work) or a 128^3 sub-patch")
    ap.add_argument("--other-workloads", default=None,
                    help="comma list of further configurations reported as extra keys (default: c3,c4,c5,prostate with --workload c2 at N=1; 'none')")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the gradient exchange: nccl (= RCCL over xGMI, one GPU per rank) or gloo "
                         "(stages device tensors through host memory; what lets N ranks share one GPU in the tests)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="ranks beyond torch.cuda.device_count() reuse the visible GPUs round-robin (gloo only: a plumbing check "
                         "of the N-rank path on a 1-GPU box, never a measurement)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N`: become the launcher -- one process per GPU under torch.distributed.run, same arguments
        import socket
        import subprocess
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
        s_.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...) or call "
                 f"`python bench.py --gpus {args.gpus}` without WORLD_SIZE in the environment (it launches the ranks itself)")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev:
        if not (args.share_gpu and args.backend == "gloo"):
            sys.exit(f"bench.py: rank {rank} has no GPU of its own ({ndev} visible, {world} ranks): one GPU per rank is the "
                     f"measurement contract (--share-gpu --backend gloo is the plumbing check)")
    dev_index = local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    use_dist = world > 1 or os.environ.get("LNN_FORCE_DP", "0") == "1"
    # stdout must carry exactly ONE line (the JSON): libraries that print to the C-level stdout (RCCL writes a version
    # banner there at communicator creation) are sent to stderr; the JSON goes to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # NO device_id: with it the RCCL communicator is created eagerly, BEFORE the engine allocates its buffers, and every
        # step then runs 6-7 % slower on this stack (ROCm 7.0 / RCCL 2.26.6; tools/dp_ab.py: 27.6 vs 25.9 ms, plain 25.9).
        # Created lazily by the first collective (the first warm-up step's gradient all-reduce) it costs nothing.
        dist.init_process_group(args.backend, rank=rank, world_size=world)

    from lifelong_nnunet_amd import native as nat
    tr, plans, ext, wl_desc, extra_cfg = build_trainer(args.workload, device, rank)

    # The iterations are enqueued the way the trainer's own epoch loop runs them (nnUNetTrainerMultiHead._run_epoch_loop): the
    # loss / gradient-norm / found-inf triple of an iteration travels to the host asynchronously and is consumed when the NEXT
    # iteration needs the loss scale -- no host synchronisation between two iterations.  Every loss is still fetched, the last one
    # inside the timed region; `config.eager_loss_fetch` is the same loop with a synchronising fetch per iteration.
    tr.defer_loss_fetch = True

    def step():
        return tr.run_iteration(tr.tr_gen, True)

    def timed(nsteps):
        torch.cuda.synchronize()
        t0_ = time.perf_counter()
        for _ in range(nsteps):
            l_ = step()
        l_ = float(l_)
        torch.cuda.synchronize()
        return time.perf_counter() - t0_, l_

    for _ in range(args.warmup):
        step()
    if tr.dp is not None:
        tr.dp.collect_timing = True          # two event records per step around the wait for the gradient exchange
    eng = list(tr.network._engines.values())[0]
    probe = None
    if rank == 0 and not args.no_roofline:
        # the three MFMA families on their heaviest layer, bracketed with HIP events on the streams they launch on,
        # INSIDE the timed steps (two event records per call: no synchronisation, no extra launches)
        probe = {"layer": heaviest_block(eng).prefix}
        eng.probe = probe
    sampler = None
    if rank == 0 and not args.no_roofline:
        try:                                              # board power during the timed steps (hwmon files, a 20 ms host thread)
            from tools.power_ceiling import Sampler
            sampler = Sampler()
            sampler.__enter__()
        except Exception:
            sampler = None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    loss = float(loss)                  # the last iteration's loss has arrived (all earlier ones were consumed on the way)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    power = None
    if sampler is not None:
        sampler.__exit__(None, None, None)
        power = sampler.summary()
    eng.probe = None
    probe_ser = None
    if probe is not None and world == 1:      # extra steps on ONE rank would issue gradient all-reduces nobody answers
        # the same bracketing with the weight gradients on the MAIN stream (the default plan runs them on a side stream next
        # to the data gradients: two MFMA-bound kernels then share the chip and each one's own duration says little)
        probe_all = {"layer": "*"}                       # every call of every layer (what tools/layer_table.py records)
        eng.probe, keep_ov = probe_all, eng.overlap_wgrad
        eng.overlap_wgrad = False
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        eng.probe, eng.overlap_wgrad = None, keep_ov
        probe_ser = {"layer": probe["layer"], "all": probe_all.get("all", [])}
        for prefix_, kind_, a_, b_ in probe_ser["all"]:
            if prefix_ == probe["layer"]:
                probe_ser.setdefault(kind_, []).append((a_, b_))
    ranks_seen, devices_seen = 1, 1
    dp_report = None
    if tr.dp is not None:
        # what the gradient exchange did in the timed steps, per rank: buckets launched from inside backward vs by finish(), and how
        # long the step's stream actually WAITED for the exchange (HIP events around the wait: the part of the all-reduce that
        # backward did not cover).  Makes an N > 1 scaling number diagnosable: LNN_DP_BUCKET_MB / LNN_DP_STREAM select the alternatives.
        st = tr.dp.stats(synchronize=True)
        dp_report = {"bucket_mb": st["bucket_mb"], "buckets": st["buckets"], "stream": st["stream"],
                     "buckets_sent_in_backward": st["buckets_sent_in_backward"], "buckets_sent_by_finish": st["buckets_sent_by_finish"],
                     "exposed_comm_ms": st.get("exposed_comm_ms"), "exposed_comm_ms_max_step": st.get("exposed_comm_ms_max")}
    rank_ms = [dt / args.steps * 1e3]
    if world > 1:
        per = [None] * world
        dist.all_gather_object(per, (dt / args.steps * 1e3, None if dp_report is None else dp_report["exposed_comm_ms"]))
        rank_ms = [p_[0] for p_ in per]
        if dp_report is not None:
            ex = [p_[1] for p_ in per if p_[1] is not None]
            dp_report["exposed_comm_ms_per_rank"] = [p_[1] for p_ in per]
            dp_report["exposed_comm_ms"] = max(ex) if ex else None       # the slowest rank's (rank 0's own value is in the list)
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)                   # every rank adds 1: what the collective library itself saw
        ranks_seen = int(ones.item())
        uu = [None] * world
        dist.all_gather_object(uu, str(torch.cuda.get_device_properties(device).uuid) if hasattr(
            torch.cuda.get_device_properties(device), "uuid") else f"index{dev_index}")
        devices_seen = len(set(uu))

    flops_patch, mac_fwd = eng.flops_per_patch()
    B = plans["batch_size"]
    patches_per_s = world * B * args.steps / dt
    out = {
        "metric": "3D patches/sec (whole node) + mean Dice vs ref, 5-level 3D Generic_UNet training step", "value": patches_per_s,
        "unit": "patches/s", "n_gpus": world, "ranks_seen": ranks_seen, "distinct_devices": devices_seen,
        "backend": (args.backend if use_dist else None), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 (fp32 accumulate, fp32 master weights)", "data": "synthetic",
        "config": {"workload": f"{wl_desc}, "
                               f"{'x'.join(map(str, plans['patch_size']))} patches, batch {B}/GPU, num_pool {plans['num_pool']}, "
                               f"base {plans['base_num_features']}, {plans['num_classes']} logits",
                   "global_batch": B * world, "parallelism": f"dp{world}", "loss": float(loss),
                   "distinct_resident_batches": len(tr.tr_gen.items) if isinstance(tr.tr_gen, ResidentBatches) else None,
                   "conv_gflop_per_patch": flops_patch / 1e9,
                   "conv_stack_tflops": patches_per_s / world * flops_patch / 1e12,
                   "conv_stack_frac_of_mfma_peak": patches_per_s / world * flops_patch / 1e12 / PEAK_MFMA_F16_TFLOPS},
    }
    out["config"].update(extra_cfg)
    # every measurement switch that is set: empty = the shipped path
    out["config"]["lnn_env"] = {k: v for k, v in sorted(os.environ.items()) if k.startswith("LNN_")}
    out["ms_per_step_rank_min"], out["ms_per_step_rank_max"] = min(rank_ms), max(rank_ms)
    if dp_report is not None:
        out["data_parallel"] = dp_report
    out["config"]["loss_fetch"] = "asynchronous: consumed by the next iteration's loss-scale decision (the trainer's epoch loop)"
    if world == 1 and not args.no_extras:
        tr.defer_loss_fetch = False
        timed(2)
        dt_e, _ = timed(args.steps)
        tr.defer_loss_fetch = True
        out["config"]["eager_loss_fetch"] = {"ms_per_step": dt_e / args.steps * 1e3, "value": B * args.steps / dt_e,
                                             "how": "run_iteration's default: one synchronising device-to-host copy per iteration"}
    if ext == "lwf":          # the fix behind a flag: every head evaluated on the training batch (one batch, one body pass)
        tr.same_batch_predictions = True
        timed(2)
        dt_sb, _ = timed(args.steps)
        tr.same_batch_predictions = False
        out["config"]["same_batch_predictions_patches_per_s"] = B * args.steps / dt_sb
        out["config"]["note"] = ("value = reference semantics (T+2 batches per iteration, one extra eval forward per head); "
                                 "conv_stack_* count the training pass only")
    if world == 1 and ext != "lwf" and not args.no_extras:
        # H2D-inclusive variant of the same loop (MH.py:606-617 includes to_cuda in the iteration)
        try:
            items = tr.tr_gen.items if isinstance(tr.tr_gen, ResidentBatches) else None
            if items:
                keep = tr.tr_gen
                tr.tr_gen = PrefetchingBatches(items, device)
                timed(2)
                dt_h, _ = timed(args.steps)
                tr.tr_gen = keep
                nbytes = sum(t.numel() * t.element_size() for t in [items[0]["data"]] + list(items[0]["target"]))
                # top-level twins of value / ms_per_step with the host-to-device copy of every batch inside the iteration
                out["value_h2d_inclusive"], out["ms_per_step_h2d_inclusive"] = B * args.steps / dt_h, dt_h / args.steps * 1e3
                out["config"]["h2d_inclusive"] = {"ms_per_step": dt_h / args.steps * 1e3, "value": B * args.steps / dt_h,
                                                  "host_bytes_per_step": nbytes,
                                                  "how": "pinned host buffers, copy of batch i+1 on a side stream during step i"}
        except Exception as e:
            out["config"]["h2d_inclusive"] = {"error": repr(e)}
    if probe is not None:
        kr = kernel_rooflines(tr)                    # the same three launches back to back on an otherwise idle chip
        fam = {"fwd": "igemm_conv_fwd", "dgrad": "igemm_conv_dgrad", "wgrad": "igemm_wgrad"}
        names = {"igemm_conv_fwd": "igemm_conv_s1_v9_kernel (stride-1 3x3x3 conv forward, z-streaming, InstanceNorm-statistics epilogue)",
                 "igemm_conv_dgrad": "igemm_conv_s1_v9_kernel (data gradient)",
                 "igemm_wgrad": "igemm_wgrad_s1_v5_kernel (stride-1 weight gradient; the rocprof top row of the step)"}
        fams = {}
        for kind, key in fam.items():
            evs, evs2 = (probe_ser or {}).get(kind, []), probe.get(kind, [])
            iso = kr["kernels"][key]
            ms_in = sum(a_.elapsed_time(b_) for a_, b_ in evs) / len(evs) if evs else None
            ms_two = sum(a_.elapsed_time(b_) for a_, b_ in evs2) / len(evs2) if evs2 else None
            fams[key] = {"kernel": names[key], "launch_ms_in_step": ms_in, "launch_ms_isolated": iso["ms"],
                         "launch_ms_in_timed_steps_two_streams": ms_two,
                         "achieved_in_step": iso["gflop"] / ms_in if ms_in else None, "achieved_isolated": iso["tflops"],
                         "frac_in_step": iso["gflop"] / ms_in / PEAK_MFMA_F16_TFLOPS if ms_in else None,
                         "frac_isolated": iso["tflops"] / PEAK_MFMA_F16_TFLOPS, "launches_timed": len(evs)}
        # the headline is the family that bounds the stack: the slowest of the three IN the step
        dom_key = min(fams, key=lambda k: fams[k]["achieved_in_step"] or fams[k]["achieved_isolated"])
        dom = fams[dom_key]
        # HBM bytes per launch / shader clock / matrix-pipe busy cycles come from committed PMC passes over this same command
        # (separate rocprofv3 runs, tools/gpu_r5_final.sh).  They describe ONE binary: each file carries the sha256 of the
        # liblnn_hip.so it was collected with and is quoted only when that is the library loaded now.
        traffic, traffic_note, clock = None, None, None
        so_sha = so_sha256()
        tj = committed_pmc("pmc_traffic.json", so_sha)
        if tj is not None:
            short = {"igemm_conv_fwd": "fwd", "igemm_conv_dgrad": "dgrad", "igemm_wgrad": "wgrad"}.get(dom_key, dom_key)
            try:
                traffic = tj[1]["kernels"][short]["hbm_bytes_per_launch_corrected"]
                traffic_note = f"profiles/{tj[0]} (so_sha256 matches the loaded library): " + tj[1].get("note", "")
            except (KeyError, TypeError) as e:
                traffic, traffic_note = None, f"profiles/{tj[0]} matches the loaded library but lacks {e!r}: traffic not quoted"
        else:
            traffic_note = "no committed PMC pass for this liblnn_hip.so (sha256 %s...): traffic not quoted" % so_sha[:12]
        cj = committed_pmc("pmc_mfma_clock.json", so_sha)
        if cj is not None:
            try:
                clock = {k: {"clock_ghz": v["clock_ghz"], "mfma_busy_frac_in_cycles": v["mfma_busy_frac_in_cycles"]}
                         for k, v in cj[1]["families"].items() if v}
                clock["source"] = f"profiles/{cj[0]} (GRBM_GUI_ACTIVE / duration; SQ_VALU_MFMA_BUSY_CYCLES / SIMD cycles)"
            except (KeyError, TypeError, AttributeError) as e:
                clock = {"error": f"profiles/{cj[0]} lacks {e!r}"}
        ach = dom["achieved_in_step"] or dom["achieved_isolated"]
        out["roofline"] = {"bound": "mfma", "kernel": dom["kernel"] + " on " + kr["layer"], "achieved": ach,
                           "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_MFMA_F16_TFLOPS,
                           "traffic": traffic, "traffic_unit": "bytes/launch (HBM, PMC)", "traffic_source": traffic_note,
                           "launch_ms": dom["launch_ms_in_step"] or dom["launch_ms_isolated"],
                           "how": "HIP events around the C-ABI call on the stream it launches on, inside training steps: "
                                  "launch_ms_in_step = steps with every kernel on one stream (what a serialised rocprof trace shows); "
                                  "launch_ms_in_timed_steps_two_streams = inside the timed steps themselves, where weight and data "
                                  "gradients of a layer run concurrently on two streams and share the chip; *_isolated = the same "
                                  "launch repeated back to back on an idle chip (the forward call includes its 2 tiny statistics "
                                  "launches)",
                           "algorithmic_gflop_per_launch": kr["kernels"][dom_key]["gflop"],
                           "families": fams, "slowest_family": dom_key, "pmc_clock_and_matrix_pipe": clock}
        # The dominant kernel over ALL its launches in the step (`frac` above is its best layer, the heaviest block): every
        # stride-1 3x3x3 launch of each family in the serialised probe steps, algorithmic FLOPs / summed duration.
        out["roofline"]["frac_best_layer"] = ach / PEAK_MFMA_F16_TFLOPS
        try:
            from lifelong_nnunet_amd.engine import ConvBlock
            blocks = {b_.prefix: b_ for b_ in eng.order if isinstance(b_, ConvBlock)}
            fam_of = {"fwd": "igemm_conv_fwd", "dgrad": "igemm_conv_dgrad", "wgrad": "igemm_wgrad"}
            tot = {k: [0.0, 0.0, 0] for k in fam_of.values()}
            for prefix_, kind_, a_, b_ in (probe_ser or {}).get("all", []):
                blk_ = blocks.get(prefix_)
                if blk_ is None or kind_ not in fam_of or not blk_.iso or blk_.stride != 1 or blk_.cin_k == 1 or blk_.ntaps != 27:
                    continue
                t_ = tot[fam_of[kind_]]
                t_[0] += 2.0 * eng.N * blk_.z.V * blk_.cin * blk_.cout * 27 / 1e9
                t_[1] += a_.elapsed_time(b_)
                t_[2] += 1
            avg = {k: {"tflops": v[0] / v[1], "frac": v[0] / v[1] / PEAK_MFMA_F16_TFLOPS, "launches": v[2], "gflop": v[0], "ms": v[1]}
                   for k, v in tot.items() if v[1] > 0}
            out["roofline"]["all_stride1_launches_by_family"] = avg
            if dom_key in avg:
                out["roofline"]["frac_kernel_avg"] = avg[dom_key]["frac"]
                out["roofline"]["frac_kernel_avg_what"] = ("every stride-1 3x3x3 launch of the dominant family (%s) in %d serialised steps: "
                                                           "algorithmic FLOPs / summed HIP-event durations" % (dom_key, args.steps))
        except Exception as e:
            out["roofline"]["frac_kernel_avg"] = None
            out["roofline"]["frac_kernel_avg_error"] = repr(e)
        # the chip is POWER-limited on random fp16 operands (profiles/r06_power_ceiling.txt: zero-filled operands run 1.39-1.53x faster
        # at 2.39 GHz, random ones at 1.6-1.8 GHz with the board pinned at its 1400 W limit, this kernel family AND the vendor GEMM)
        out["roofline"]["power_w_in_timed_steps"] = None if power is None else power.get("power_w")
        fam_short = {"igemm_conv_fwd": "fwd", "igemm_conv_dgrad": "dgrad", "igemm_wgrad": "wgrad"}.get(dom_key)
        fam_clock = (clock or {}).get(fam_short)
        out["roofline"]["sustained_clock_ghz"] = fam_clock.get("clock_ghz") if isinstance(fam_clock, dict) else None
        out["roofline"]["power_ceiling"] = ("profiles/r06_power_ceiling.txt: on random fp16 data the board sits at its 1400 W limit and the shader "
                                            "clock at 1.6-1.8 GHz (2.39 GHz on zero-filled operands, +39..53 % throughput for this family and for "
                                            "hipBLASLt alike); reachable fraction of the 2.5 PFLOP/s datasheet peak ~= matrix-pipe busy fraction x "
                                            "1.7 / 2.4 -- the vendor GEMM reaches 0.51")
        try:
            ref = library_gemm_reference()
            out["roofline"]["library_gemm_fp16_reference"] = ref
            out["roofline"]["library_gemm_frac"] = ref["tflops"] / PEAK_MFMA_F16_TFLOPS
            out["roofline"]["achieved_vs_library_gemm"] = ach / ref["tflops"]
        except Exception as e:
            out["roofline"]["library_gemm_fp16_reference"] = {"error": repr(e)}
        try:
            out["regulariser_kernels"] = regulariser_rooflines(tr, plans)
        except Exception as e:
            out["regulariser_kernels"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            batch = tr.tr_gen.items[0] if isinstance(tr.tr_gen, ResidentBatches) else next(tr.tr_gen)
            out["cpu_baseline"], out["parity"] = cpu_baseline_and_parity(tr, plans, flops_patch, batch, args.cpu_sample)
        except Exception as e:      # the GPU numbers above must still be reported
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    others = args.other_workloads
    if others is None:
        others = "c3,c4,c5,prostate" if (args.workload == "c2" and world == 1 and not args.no_roofline) else "none"
    if rank == 0 and world == 1 and others != "none":
        del tr, eng
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        out["other_workloads"] = {}
        for wname in others.split(","):
            try:
                out["other_workloads"][wname] = other_workload(wname, args, device, rank)
            except Exception as e:
                out["other_workloads"][wname] = {"error": repr(e)}
            gc.collect()
            torch.cuda.empty_cache()
    if rank == 0:
        info = nat.device_info()
        out["device"] = info
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        if world > 1:
            dist.barrier()          # rank 0 may still be timing the roofline kernels: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
