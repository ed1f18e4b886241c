#!/bin/bash
# round 3, call O: z-streaming stride-2 / transposed-conv weight gradient (igemm_wgrad_s2s): tests, per-layer A/B, step time
TAG=${1:-r3o}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu --timeout=120 -k "wgrad or convT or determin" > $OUT/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest.log | tail -12
L=enc1.0s2,enc2.0s2,enc3.0s2,enc4.0s2,up4,up3,up2
for x in 0 1; do echo "== LNN_WGRAD_S2S=$x"; LNN_WGRAD_S2S=$x timeout 200 python tools/kbench.py --layers $L --which wgrad --iters 20 2>&1 | grep -v amdgpu.ids | tail -7; done | tee $OUT/kbench_s2s.txt
timeout 600 python -m pytest tests/test_training_gpu.py tests/test_fullsize_gpu.py -q -m gpu --timeout=300 > $OUT/pytest2.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest2.log | tail -8
for v in "LNN_WGRAD_S2S=0" "LNN_WGRAD_S2S=1"; do
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --other-workloads none 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'])"
done | tee $OUT/step_ab.txt
