#!/bin/bash
# last lease of round 4: the PMC passes on the final binary (re-stamped), the loss-kernel tests, the default bench line
TAG=${1:-r4last}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_training_gpu.py -q -k "dice or c1_iterations or golden" 2>&1 | tail -2
timeout 100 python tools/microbench_reductions.py 2>&1 | tail -1
bash tools/gpu_r4_pmc.sh $TAG 2>&1 | grep -v "g=" | tail -8
cp $OUT/pmc_traffic.json profiles/r04_pmc_traffic.json; cp $OUT/pmc_mfma_clock.json profiles/r04_pmc_mfma_clock.json
timeout 900 python bench.py > $OUT/bench_c2.json 2> $OUT/bench.err; tail -2 $OUT/bench.err | cut -c1-300
python - <<PY
import json
d=json.load(open("$OUT/bench_c2.json"))
print(d["value"], d["ms_per_step"], round(d["roofline"]["frac"],3), d["roofline"].get("traffic"))
print({k:(round(v["GBps"]),round(v["ms"],3)) for k,v in d.get("regulariser_kernels",{}).get("kernels",{}).items() if "GBps" in v})
print(d["parity"]["loss_rel_err"], {k:(v.get("value"), (v.get("parity") or {}).get("gates")) for k,v in d["other_workloads"].items()})
PY
