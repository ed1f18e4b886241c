#!/bin/bash
# round 3, call B: whole GPU suite, new bench line, serialised kernel table
TAG=${1:-r3b}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x > $OUT/pytest.log 2>&1; tail -8 $OUT/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err | cut -c1-300
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d["value"], d["ms_per_step"], d["metric"])
print(json.dumps(d.get("roofline",{}).get("families"),indent=0)[:1500])
print(d.get("parity")); print(d.get("cpu_baseline"))
print({k:(v.get("value"),v.get("ms_per_step"),v.get("error")) for k,v in d.get("other_workloads",{}).items()})
PY
d2=/tmp/prof2_$TAG; rm -rf $d2
(cd /tmp && LNN_NO_WGRAD_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace -d $d2 -o r -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/prof2_bench.json 2> $OUT/prof2.err)
python tools/rocpd_stats.py $(find $d2 -name "*.db" | head -1) > $OUT/kernel_stats_serialized.txt 2>&1
head -45 $OUT/kernel_stats_serialized.txt | cut -c1-170
