#!/bin/bash
# round-4 lease D: the plan / generic-kernel / fp32 / LwF tests that lease C's -x cut off, and the reduction-kernel variants
TAG=${1:-r4d}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_plans_gpu.py tests/test_gen_gpu.py tests/test_fp32_parity_gpu.py tests/test_kernels_gpu.py -q -s --timeout=300 > $OUT/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed|Error" $OUT/pytest.log | tail -12; grep -E "rel |relative error|same-batch" $OUT/pytest.log | tail -40
for v in 0 1 2 3 6 7 10 11; do LNN_GRADNORM_VARIANT=$v LNN_DCE_BLOCKS=$((256 << (v % 4))) timeout 120 python tools/microbench_reductions.py 2>/dev/null | tail -1; done | tee $OUT/microbench.txt
