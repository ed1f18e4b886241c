#!/bin/bash
# round 3, call S: next-tile decode inside the MFMA loop (wgrad v5, s2s): phases, per-layer timing, tests, step
TAG=${1:-r3s}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 120 python tools/kbench.py --layers dec4.0,enc0.1,enc1.1,dec3.0,dec2.0,enc1.0s2,enc2.0s2,up3 --which wgrad --iters 20 --wgrad-phases 2>&1 | grep -v amdgpu.ids | tail -14 | tee $OUT/wgrad.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_training_gpu.py -q -m gpu --timeout=120 -k "wgrad or convT or determin or step or train" 2>&1 | tail -2
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --other-workloads none 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], d['value'])"
