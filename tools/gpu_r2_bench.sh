#!/bin/bash
# bench lines of all workloads.  usage: bash tools/gpu_r2_bench.sh <tag>
TAG=${1:-e}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
for w in c2 c3 c4 c5; do
  extra=""; [ $w != c2 ] && extra="--no-cpu-baseline"
  timeout 900 python bench.py --workload $w --steps 10 --warmup 3 $extra > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  echo "== $w rc=$?"; tail -3 $OUT/bench_$w.err | cut -c1-300; python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$w.json"))
    print(d["value"], d["ms_per_step"], {k:v for k,v in d["config"].items() if k not in ("workload",)})
    print({k:(round(v["GBps"]),round(v["ms"],3)) for k,v in d.get("regulariser_kernels",{}).get("kernels",{}).items() if "GBps" in v})
    print(d.get("roofline",{}).get("achieved"), d.get("roofline",{}).get("other_kernels"))
except Exception as e: print("ERR",e)
PY
done
