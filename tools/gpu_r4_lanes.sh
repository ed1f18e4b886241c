#!/bin/bash
# two-lane mode (engine._fork): parity in lane mode, then the C2 step for several CU budgets against the single-lane step, same box
TAG=${1:-r4lanes}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
LNN_SAMPLE_LANES=1 timeout 600 python -m pytest tests/test_training_gpu.py -q --timeout=300 -k "not deterministic" > $OUT/pytest_lanes.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest_lanes.log | tail -8
timeout 600 python -m pytest tests/test_training_gpu.py tests/test_kernels_gpu.py -q --timeout=300 > $OUT/pytest_single.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest_single.log | tail -8
run() { # label, env...
  label=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --other-workloads none > $OUT/bench_$label.json 2> $OUT/bench_$label.err
  python -c "import json;d=json.load(open('$OUT/bench_$label.json'));print('$label',round(d['value'],2),round(d['ms_per_step'],3),d['config']['loss'])" 2>&1 | tee -a $OUT/summary.txt
}
run single LNN_SAMPLE_LANES=0
run lanes192 LNN_SAMPLE_LANES=1 LNN_LANE_CUS=192
run lanes208 LNN_SAMPLE_LANES=1 LNN_LANE_CUS=208
run lanes240 LNN_SAMPLE_LANES=1 LNN_LANE_CUS=240
run lanes256 LNN_SAMPLE_LANES=1 LNN_LANE_CUS=256
run lanes192_nostagger LNN_SAMPLE_LANES=1 LNN_LANE_CUS=192 LNN_LANE_STAGGER=0
run lanes160 LNN_SAMPLE_LANES=1 LNN_LANE_CUS=160
run single_again LNN_SAMPLE_LANES=0
