#!/bin/bash
# round 3, call E: 2x2x2 streaming kernel + kl_logits: tests and timings (short)
TAG=${1:-r3e}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu --timeout=120 -k "streaming or kl_logits or convT" > $OUT/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest.log | tail -8
timeout 200 python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
from lifelong_nnunet_amd import native as nat
dev = "cuda:0"
def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for (N, C, K, D, H, W) in [(2, 64, 32, 80, 96, 80), (2, 128, 64, 40, 48, 40)]:
    dy = (torch.randn((N, 2 * D, 2 * H, 2 * W, K), device=dev) * 0.5).half()
    dx = torch.empty((N, D, H, W, C), dtype=torch.float16, device=dev)
    wd = torch.randn(nat.query("lnn_packed_weight_elems", 8, C, K), device=dev).half()
    for which in (1, 0):
        nat.lib().lnn_debug_force_down2_kernel(which)
        ms = t(lambda: nat.call("lnn_convT3d_k2s2_dgrad", dy, K, wd, dx, C, N, D, H, W, C, K, 0))
        gb = (dy.numel() + dx.numel()) * 2 / 1e9
        print(f"convT dgrad {K}->{C} @{D}x{H}x{W}: kernel {'streaming' if which else 'tile'} {ms:.3f} ms  {gb / ms:.2f} TB/s", flush=True)
nat.lib().lnn_debug_force_down2_kernel(-1)
B, K, V = 2, 3, 160 * 160 * 160
lg, lt = torch.randn((B, K, V), device=dev), torch.randn((B, K, V), device=dev)
out, ws = torch.zeros(1, device=dev), torch.zeros(nat.query("lnn_kl_logits_ws_doubles", B), dtype=torch.float64, device=dev)
ms = t(lambda: nat.call("lnn_kl_logits", lg, lt, B, K, V, 2.0, out, ws))
print(f"kl_logits {B}x{K}x{V}: {ms:.3f} ms {2 * 4 * B * K * V / ms / 1e9:.2f} TB/s")
PY
