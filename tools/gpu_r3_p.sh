#!/bin/bash
# round 3, call P/Q: igemm_wgrad_s2s variants: per-layer timing + the weight-gradient tests
TAG=${1:-r3p}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
L=enc1.0s2,enc2.0s2,enc3.0s2,enc4.0s2,up4,up3,up2
for x in 1 0 1; do echo "== LNN_WGRAD_S2S=$x"; LNN_WGRAD_S2S=$x timeout 200 python tools/kbench.py --layers $L --which wgrad --iters 30 2>&1 | grep -v amdgpu.ids | tail -7; done | tee $OUT/kbench_s2s.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu --timeout=120 -k "wgrad or convT or determin" 2>&1 | tail -2
