#!/bin/bash
# round 6, call B: full GPU suite + default bench + per-layer table with the macro-tile kernel in the automatic selection
TAG=${1:-r6b}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
sha256sum lifelong-nnunet_amd/csrc/liblnn_hip.so | tee $OUT/so_sha256.txt
timeout 1500 python -m pytest tests -q -m gpu --timeout=600 -x > $OUT/pytest_gpu.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest_gpu.log | tail -8
timeout 600 python bench.py --other-workloads none --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/bench.err; tail -2 $OUT/bench.err | cut -c1-300
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_c2.json"))
    print(d["value"], d["ms_per_step"], d.get("ms_per_step_h2d_inclusive"), round(d["roofline"]["frac"],3), d["config"]["conv_stack_frac_of_mfma_peak"])
    print(d.get("parity"))
except Exception as e: print("ERR", e)
PY
timeout 300 python tools/layer_table.py --steps 6 > $OUT/layer_table.txt 2> $OUT/layer_table.err; tail -22 $OUT/layer_table.txt
