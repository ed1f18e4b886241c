#!/bin/bash
# round 3, call U: side-stream weight gradients enqueued behind their layer's data gradient (LNN_WGRAD_LAG=1) vs next to it
TAG=${1:-r3u}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
for v in "LNN_WGRAD_LAG=0" "LNN_WGRAD_LAG=1" "LNN_WGRAD_LAG=0" "LNN_WGRAD_LAG=1"; do
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --other-workloads none 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'], d['config']['loss'])"
done | tee $OUT/step_ab.txt
