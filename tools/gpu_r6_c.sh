#!/bin/bash
# round 6, call C: the new GPU tests (side-data round trips, DP report, macro-tile after the variant removal), sliding-window error print
TAG=${1:-r6c}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_fp32_parity_gpu.py tests/test_bench_launch_gpu.py -q -m gpu -x --timeout=600 -k "restored or lwf_restore or bench" > $OUT/pytest_new.log 2>&1; tail -15 $OUT/pytest_new.log
timeout 600 python -m pytest tests/test_training_gpu.py -q -m gpu -s -k "sliding_window_inference" > $OUT/pytest_sw.log 2>&1; grep -E "max\|dp\||passed|failed" $OUT/pytest_sw.log | tail
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "deep_layers or macro_tile or every_stride1 or cat_ops or splitk_small" --timeout=600 > $OUT/pytest_mt.log 2>&1; tail -3 $OUT/pytest_mt.log
timeout 900 python -m pytest tests -q -m gpu --timeout=600 > $OUT/pytest_gpu.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest_gpu.log | tail -8
