#!/usr/bin/env python
"""Host-side cost of ENQUEUEING one C2 training step: the trainer runs with every C-ABI launch replaced by a no-op (torch's own
small ops still run), so what is timed is Python + ctypes argument marshalling + torch glue -- the time the host needs per step
whatever the GPU does.  A step is GPU-bound only while this stays below the GPU's step time.
    python tools/host_overhead.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lifelong_nnunet_amd import native as nat

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
tr, plans, ext, desc, _ = bench.build_trainer("c2", dev, 0)
for _ in range(3):
    tr.run_iteration(tr.tr_gen, True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    tr.run_iteration(tr.tr_gen, True)
torch.cuda.synchronize()
real = (time.perf_counter() - t0) / 10
count = [0]
orig = nat.call


def stub(name, *args):
    count[0] += 1
    conv = [a.data_ptr() if hasattr(a, "data_ptr") else a for a in args]      # keep the marshalling cost
    nat.stream_handle()


nat.call = stub
for _ in range(2):
    tr.run_iteration(tr.tr_gen, True)
torch.cuda.synchronize()
count[0] = 0
t0 = time.perf_counter()
for _ in range(10):
    tr.run_iteration(tr.tr_gen, True)
torch.cuda.synchronize()
host = (time.perf_counter() - t0) / 10
nat.call = orig
print(f"real step {real * 1e3:.2f} ms | host-only (launches stubbed) {host * 1e3:.2f} ms "
      f"for {count[0] / 10:.0f} C-ABI calls per step = {host / (count[0] / 10) * 1e6:.1f} us per call")
