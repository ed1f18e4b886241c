#!/bin/bash
# round-4 lease C: plan-driven networks (anisotropic / multi-channel), generic kernels with the measured dispatch, fused gradnorm,
# restructured Dice+CE forward: parity tests, full suite, default bench line
TAG=${1:-r4c}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 600 python -m pytest tests/test_plans_gpu.py tests/test_gen_gpu.py -q --timeout=300 -x > $OUT/pytest_plans.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed|Error|rel " $OUT/pytest_plans.log | tail -12
timeout 900 python -m pytest tests -q -m gpu --timeout=300 --deselect tests/test_gen_gpu.py --deselect tests/test_plans_gpu.py > $OUT/pytest_gpu.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest_gpu.log | tail -8
timeout 900 python bench.py --other-workloads none > $OUT/bench_c2.json 2> $OUT/bench.err; tail -2 $OUT/bench.err | cut -c1-300
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_c2.json"))
    print(d["value"], d["ms_per_step"], d["roofline"]["slowest_family"], round(d["roofline"]["frac"],3))
    print(d.get("parity"))
    print({k:(round(v["GBps"]),round(v["ms"],3)) for k,v in d.get("regulariser_kernels",{}).get("kernels",{}).items() if "GBps" in v})
except Exception as e: print("ERR", e)
PY
