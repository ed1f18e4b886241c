#!/bin/bash
# round 3, call F: folded first-layer backward, staged seg-forward stores, faster finalize / pack kernels: tests, bench, profile
TAG=${1:-r3g}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests -q -m gpu --timeout=300 > $OUT/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest.log | tail -15
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --other-workloads none > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err | cut -c1-300
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench.json"))
    print(d["value"], d["ms_per_step"])
    print({k:(round(v["launch_ms_in_step"],3), round(v["launch_ms_isolated"],3), round(v["launch_ms_in_timed_steps_two_streams"],3)) for k,v in d["roofline"]["families"].items()})
    print({k:(round(v["GBps"]),round(v["ms"],3)) for k,v in d.get("regulariser_kernels",{}).get("kernels",{}).items() if "GBps" in v})
except Exception as e: print("ERR", e)
PY
d2=/tmp/prof2_$TAG; rm -rf $d2
(cd /tmp && LNN_NO_WGRAD_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace -d $d2 -o r -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/prof2_bench.json 2> $OUT/prof2.err)
python tools/rocpd_stats.py $(find $d2 -name "*.db" | head -1) > $OUT/kernel_stats_serialized.txt 2>&1
head -50 $OUT/kernel_stats_serialized.txt | cut -c1-170
