#!/usr/bin/env python
"""Per-layer, per-direction time of the C2 training step: HIP events around EVERY conv / normalisation call of the engine
(engine.probe = {"layer": "*"}), weight gradients on the main stream so that every duration is the call's own.
    python tools/layer_table.py [--steps 6] [--workload c2] > gpurun_out/layer_table.txt
Columns: ms per call (mean over the steps), algorithmic GFLOP of the call, TFLOP/s, share of the step's kernel time."""
import argparse
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--workload", default="c2")
    a = ap.parse_args()
    import torch
    import bench
    from lifelong_nnunet_amd.engine import ConvBlock, UpBlock
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    tr, plans, ext, desc, _ = bench.build_trainer(a.workload, dev, 0)
    for _ in range(3):
        tr.run_iteration(tr.tr_gen, True)
    eng = list(tr.network._engines.values())[0]
    eng.overlap_wgrad = False
    probe = {"layer": "*"}
    eng.probe = probe
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        tr.run_iteration(tr.tr_gen, True)
    e1.record()
    torch.cuda.synchronize()
    eng.probe = None
    step_ms = e0.elapsed_time(e1) / a.steps
    acc = defaultdict(list)
    for prefix, kind, a0, a1 in probe["all"]:
        acc[(prefix, kind)].append(a0.elapsed_time(a1))
    items = {it.prefix: it for it in eng.order if isinstance(it, (ConvBlock, UpBlock))}
    N = eng.N
    rows, fam = [], defaultdict(float)
    for (prefix, kind), v in acc.items():
        it = items[prefix]
        ms = sum(v) / len(v)
        if isinstance(it, ConvBlock):
            fl = 2.0 * N * it.z.V * it.cin * it.cout * it.ntaps
            geo = f"s{it.stride}" if it.iso else "k" + "".join(map(str, it.kernel)) + "/s" + "".join(map(str, it.strides))
            shape = f"{it.cin}->{it.cout} {geo} @{'x'.join(map(str, it.z.dims))}"
            level = eng.dims.index(tuple(it.z.dims))
        else:
            kt = tuple(it.strides)          # kernel == stride
            fl = 2.0 * N * it.x.V * it.cin * it.cout * it.ntaps
            shape = f"{it.cin}->{it.cout} convT{''.join(map(str, kt))} @{'x'.join(map(str, it.x.dims))}"
            level = eng.dims.index(tuple(it.x.dims)) - 1
        if kind.startswith("in_"):
            fl = 0.0
        rows.append((prefix, kind, shape, level, ms, fl))
        key = ("norm" if kind.startswith("in_") else ("s2/convT" if (isinstance(it, UpBlock) or it.stride == 2) else
                                                     ("first" if it.first else ("s1" if it.iso else "aniso")))) + f" L{level}"
        fam[key] += ms
    tot = sum(r[4] for r in rows)
    print(f"# {desc}; step {step_ms:.2f} ms with probes (weight gradients on the main stream); probed calls {tot:.2f} ms")
    print(f"{'layer':42s} {'kind':7s} {'shape':34s} {'ms':>8s} {'GFLOP':>9s} {'TF/s':>7s} {'%':>5s}")
    for prefix, kind, shape, level, ms, fl in sorted(rows, key=lambda r: -r[4]):
        print(f"{prefix:42s} {kind:7s} {shape:34s} {ms:8.3f} {fl / 1e9:9.1f} {fl / ms / 1e9 if fl else 0:7.0f} {100 * ms / tot:5.1f}")
    print("\n# by family and level (ms per step)")
    for k in sorted(fam):
        print(f"{k:16s} {fam[k]:7.3f}")


if __name__ == "__main__":
    main()
