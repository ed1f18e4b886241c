#!/usr/bin/env python
"""Benchmark of the hot path: 3-D U-Net training iterations (nnUNetTrainerSequential.run_iteration) on
synthetic 160x192x160 patches -- BASELINE.json configs[1] -- one process per GPU.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one complete optimisation step (H2D-free: the patches are resident in HBM): forward, deep-supervised
Dice+CE, scaled backward, gradient all-reduce (N > 1), clip 12, SGD-Nesterov, head re-sync, loss fetch.
Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").
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

WORKLOADS = {
    "c2": {"patch_size": (160, 192, 160), "batch_size": 2, "num_pool": 5, "base_num_features": 32,
           "num_classes": 3, "num_input_channels": 1, "synthetic_period": 1},
    "c1": {"patch_size": (40, 56, 40), "batch_size": 2, "num_pool": 3, "base_num_features": 32,
           "num_classes": 3, "num_input_channels": 1, "synthetic_period": 1},
}


class ResidentBatches:
    """The reference's data dict, with tensors already in HBM (inputs resident when the timed region starts)."""

    def __init__(self, gen, device, n=1):
        self.items = []
        for _ in range(n):
            d = next(gen)
            self.items.append({"data": d["data"].to(device), "target": [t.to(device) for t in d["target"]], "keys": d["keys"]})
        self.i = 0

    def __iter__(self):
        return self

    def __next__(self):
        self.i += 1
        return self.items[self.i % len(self.items)]


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


def cpu_baseline(plans, flops_full):
    """The oracle (pure PyTorch CPU fp32 restatement of the reference's step) on the GPU box's host cores, on a
    bounded sample: the SAME 5-level network on a 128x128x128 sub-patch (43 % of the voxels), B=1, one warm-up and
    one timed iteration; converted to full-size patches/s by the voxel ratio."""
    import torch
    from oracle import losses as olosses, train as otrain
    from oracle.unet import OracleGenericUNet
    from lifelong_nnunet_amd.synthetic import make_patch_batch
    cores = min(os.cpu_count() or 1, 32)     # oneDNN conv3d stops scaling (and thrashes) far below 256 threads
    torch.set_num_threads(cores)
    q = 2 ** plans["num_pool"]
    sub = tuple(max(2 * q, min(128, p // q * q)) for p in plans["patch_size"])      # divisible by 2^num_pool, ~10 s of CPU work
    torch.manual_seed(0)
    net = OracleGenericUNet(1, plans["base_num_features"], plans["num_classes"], plans["num_pool"])
    opt = otrain.make_optimizer(net)
    w = olosses.ds_loss_weights(plans["num_pool"])
    data, tgts = make_patch_batch(1, sub, plans["num_pool"], seed=1)
    small = make_patch_batch(1, tuple(2 ** (plans["num_pool"] + 1) for _ in sub), plans["num_pool"], seed=2)
    otrain.run_iteration(net, opt, small[0], small[1], w)        # warm-up (thread pools, oneDNN primitives)
    t0 = time.time()
    otrain.run_iteration(net, opt, data, tgts, w)
    dt = time.time() - t0
    ratio = 1.0
    for a, b in zip(sub, plans["patch_size"]):
        ratio *= a / b
    cpu_name = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_name = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": ratio / dt, "unit": "patches/s", "cores": cores, "kind": "port", "cpu": cpu_name,
            "sample": f"oracle.train.run_iteration, same {plans['num_pool']}-level U-Net, ONE {sub[0]}x{sub[1]}x{sub[2]} "
                      f"patch (B=1, {ratio:.4f} of the voxels of a {'x'.join(map(str, plans['patch_size']))} patch) in "
                      f"{dt:.1f} s, scaled by the voxel ratio",
            "gflops": flops_full * ratio / dt / 1e9}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
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
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from lifelong_nnunet_amd import get_trainer_class, native as nat
    plans = dict(WORKLOADS[args.workload])
    Trainer = get_trainer_class("sequential")

    def provider(task, split, p):
        from lifelong_nnunet_amd.training.network_training.multihead.nnUNetTrainerMultiHead import default_data_provider
        return ResidentBatches(default_data_provider(task, split, p, seed=12345 + 7919 * rank), device)

    tr = Trainer("seg_outputs", "synthetic_task_A", plans=plans, data_provider=provider, device=device, fold=0)
    tr.initialize(True, num_epochs=1000)
    tr.network.train()

    def step():
        return tr.run_iteration(tr.tr_gen, True)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)

    eng = list(tr.network._engines.values())[0]
    flops_patch, mac_fwd = eng.flops_per_patch()
    B = plans["batch_size"]
    patches_per_s = world * B * args.steps / dt
    out = {
        "metric": "3D patches/sec (whole node), 5-level 3D Generic_UNet training step", "value": patches_per_s,
        "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 (fp32 accumulate, fp32 master weights)", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: nnUNetTrainerSequential.run_iteration, "
                               f"{'x'.join(map(str, plans['patch_size']))} patches, batch {B}/GPU, num_pool {plans['num_pool']}, "
                               f"base {plans['base_num_features']}, {plans['num_classes']} logits",
                   "global_batch": B * world, "parallelism": f"dp{world}", "loss": float(loss),
                   "conv_gflop_per_patch": flops_patch / 1e9,
                   "conv_stack_tflops": patches_per_s / world * flops_patch / 1e12,
                   "conv_stack_frac_of_mfma_peak": patches_per_s / world * flops_patch / 1e12 / PEAK_MFMA_F16_TFLOPS},
    }
    if rank == 0 and not args.no_roofline:
        kr = kernel_rooflines(tr)
        dom = min(kr["kernels"].items(), key=lambda kv: kv[1]["tflops"])     # the slowest family bounds the stack
        fwd = kr["kernels"]["igemm_conv_fwd"]
        traffic, traffic_note = None, None
        try:      # HBM bytes per launch of the same kernel/launch from the committed PMC passes (separate rocprofv3 runs)
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            traffic, traffic_note = tj["hbm_bytes_per_launch_corrected"], "profiles/r01_pmc_traffic.json: " + tj["note"]
        except Exception:
            pass
        out["roofline"] = {"bound": "mfma", "kernel": "igemm_conv_s1_v5_kernel (stride-1 3x3x3 conv fwd) on " + kr["layer"], "achieved": fwd["tflops"],
                           "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s", "frac": fwd["tflops"] / PEAK_MFMA_F16_TFLOPS,
                           "traffic": traffic, "traffic_unit": "bytes/launch (HBM, PMC)", "traffic_source": traffic_note,
                           "launch_ms": fwd["ms"], "algorithmic_gflop_per_launch": fwd["gflop"],
                           "other_kernels": {k: {"achieved": v["tflops"], "frac": v["tflops"] / PEAK_MFMA_F16_TFLOPS,
                                                 "launch_ms": v["ms"]} for k, v in kr["kernels"].items()},
                           "slowest_family": dom[0]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(plans, flops_patch)
        except Exception as e:      # the GPU numbers above must still be reported
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
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
