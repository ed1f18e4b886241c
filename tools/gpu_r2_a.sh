#!/bin/bash
# round-2 GPU session A: v9 parity + A/B timing against v5/v8
set -x
mkdir -p gpurun_out/a
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "stride1_conv_kernel_variant or v9 or cat_ops" > gpurun_out/a/pytest_v9.log 2>&1
tail -15 gpurun_out/a/pytest_v9.log
for v in 0 1; do
  LNN_CONV_V9=$v timeout 300 python tools/kbench.py --layers enc0.1,dec4.0cat,dec4.0,enc1.1 --which fwd,dgrad --iters 5 > gpurun_out/a/kbench_v9_$v.log 2>&1
  cat gpurun_out/a/kbench_v9_$v.log
done
timeout 300 python tools/kbench.py --layers enc0.1,dec4.0,enc1.1 --which fwd --iters 1 --check 5,9 > gpurun_out/a/check.log 2>&1
cat gpurun_out/a/check.log
