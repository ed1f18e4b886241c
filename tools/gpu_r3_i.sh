#!/bin/bash
# round 3, call I: grouped XCD-aware block order of the weight-gradient kernels: tests, per-layer A/B, step A/B (alternating), forced DP
TAG=${1:-r3i}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu --timeout=120 -k "wgrad or determin" > $OUT/pytest_kernels.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest_kernels.log | tail -8
L=enc2.0s2,enc3.0s2,enc4.0s2,enc1.0s2,up3,up2,dec4.0,dec3.0,enc1.1,dec2.0,enc2.1,enc3.1,dec1.0,enc4.1,dec0.0,enc5.1
for x in 0 1; do echo "== LNN_WGRAD_XCD=$x"; LNN_WGRAD_XCD=$x timeout 200 python tools/kbench.py --layers $L --which wgrad --iters 10 2>&1 | grep -v amdgpu.ids | tail -16; done | tee $OUT/kbench_xcd.txt
for v in "LNN_WGRAD_XCD=0" "LNN_WGRAD_XCD=1" "LNN_FORCE_DP=1" "LNN_WGRAD_XCD=0" "LNN_WGRAD_XCD=1" "LNN_FORCE_DP=1"; do
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --other-workloads none 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'])"
done | tee $OUT/step_ab.txt
