#!/bin/bash
# round-4 lease A: full GPU suite on the cleaned tree, per-layer table of the C2 step, default bench line (new parity gates)
TAG=${1:-r4a}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests -q -m gpu --timeout=300 -x > $OUT/pytest_gpu.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest_gpu.log | tail -8
timeout 300 python tools/layer_table.py --steps 6 > $OUT/layer_table.txt 2> $OUT/layer_table.err; tail -3 $OUT/layer_table.err; head -40 $OUT/layer_table.txt | cut -c1-150
timeout 900 python bench.py > $OUT/bench_c2.json 2> $OUT/bench.err; tail -2 $OUT/bench.err | cut -c1-300
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_c2.json"))
    print(d["value"], d["ms_per_step"], d["roofline"]["slowest_family"], round(d["roofline"]["frac"],3), d["roofline"]["traffic_source"][:80])
    print(d.get("cpu_baseline")); print(d.get("parity"))
    print({k:(v.get("value"), v.get("parity")) if isinstance(v,dict) else v for k,v in d.get("other_workloads",{}).items()})
    print({k:(round(v["GBps"]),round(v["ms"],3)) for k,v in d.get("regulariser_kernels",{}).get("kernels",{}).items() if "GBps" in v})
except Exception as e: print("ERR", e)
PY
