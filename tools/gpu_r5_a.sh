#!/bin/bash
# round-5 lease A: sanity tests of the lane-free engine, lazy top-level z A/B, deferred weight-gradient schedules A/B, step timeline
TAG=${1:-r5a}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 600 python -m pytest tests/test_training_gpu.py tests/test_cl_gpu.py tests/test_fullsize_gpu.py -q -m gpu -x --timeout=300 > $OUT/pytest_a.log 2>&1; tail -3 $OUT/pytest_a.log
b() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$name.json")); print("$name", round(d["ms_per_step"], 3), "ms", round(d["value"], 2), "patches/s", "loss", d["config"]["loss"])
except Exception as e:
    print("$name ERR", e)
PY
}
b base0 LNN_NO_LAZY_TOP_Z=1
b lazy0 X=1
b defer_0_3 LNN_WGRAD_DEFER=0,3
b defer_1_3 LNN_WGRAD_DEFER=1,3
b defer_1_2 LNN_WGRAD_DEFER=1,2
b defer_1_4 LNN_WGRAD_DEFER=1,4
b defer_0_2 LNN_WGRAD_DEFER=0,2
b defer_2_4 LNN_WGRAD_DEFER=2,4
b defer_1_3_192 LNN_WGRAD_DEFER=1,3,192
b defer_1_3_128 LNN_WGRAD_DEFER=1,3,128
b defer_0_3_128 LNN_WGRAD_DEFER=0,3,128
b base1 LNN_NO_LAZY_TOP_Z=1
b lazy1 X=1
for v in lazy defer_1_3; do
  d=/tmp/prof_$v; rm -rf $d
  e="X=1"; [ $v = defer_1_3 ] && e="LNN_WGRAD_DEFER=1,3"
  (cd /tmp && env $e timeout 300 rocprofv3 --kernel-trace -d $d -o r -- python $OLDPWD/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/prof_$v.json 2> $OUT/prof_$v.err)
  db=$(find $d -name "*.db" | head -1)
  python tools/step_timeline.py $db --step -2 > $OUT/timeline_$v.txt 2>&1; tail -4 $OUT/timeline_$v.txt
  python tools/rocpd_stats.py $db > $OUT/kernel_stats_$v.txt 2>&1
done
