#!/bin/bash
# round 3, call T: reproducibility test repeated (is the deterministic mode still bit-exact with the new stride-2 wgrad?), wgrad tests
TAG=${1:-r3t}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
for i in 1 2 3; do timeout 300 python -m pytest tests/test_training_gpu.py -q -m gpu --timeout=120 -k "reproducible" > $OUT/repro_$i.log 2>&1; tail -1 $OUT/repro_$i.log; done
grep -B5 -A25 "Error\|assert" $OUT/repro_1.log | head -80
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_training_gpu.py -q -m gpu --timeout=120 2>&1 | tail -2
