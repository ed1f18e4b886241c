#!/bin/bash
# round-4 evidence, part 1: full GPU suite, smoke, the default bench line (CPU baseline + parity blocks + other workloads),
# rocprofv3 --kernel-trace --stats of the bench command (two streams) and serialised, the per-layer table of the step
TAG=${1:-r4final}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 1200 python -m pytest tests -q -m gpu --timeout=300 > $OUT/pytest_gpu.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest_gpu.log | tail -8
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > $OUT/bench_c2.json 2> $OUT/bench.err; tail -2 $OUT/bench.err | cut -c1-300
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_c2.json"))
    print(d["value"], d["ms_per_step"], d["roofline"]["slowest_family"], round(d["roofline"]["frac"],3), d["config"]["conv_stack_frac_of_mfma_peak"])
    print({k:(round(v["launch_ms_in_step"],3), round(v["launch_ms_isolated"],3), round(v["launch_ms_in_timed_steps_two_streams"],3)) for k,v in d["roofline"]["families"].items()})
    print(d.get("cpu_baseline")); print(d.get("parity"))
    print({k:(v.get("value"), (v.get("parity") or {}).get("gates")) if isinstance(v,dict) else v for k,v in d.get("other_workloads",{}).items()})
    print({k:(round(v["GBps"]),round(v["ms"],3)) for k,v in d.get("regulariser_kernels",{}).get("kernels",{}).items() if "GBps" in v})
except Exception as e: print("ERR", e)
PY
d=/tmp/prof_$TAG; rm -rf $d
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $d -o r -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --other-workloads none > $OUT/prof_bench.json 2> $OUT/prof.err)
python tools/rocpd_stats.py $(find $d -name "*.db" | head -1) > $OUT/kernel_stats.txt 2>&1; head -8 $OUT/kernel_stats.txt | cut -c1-170
cp $(find $d -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats.csv 2>/dev/null
d2=/tmp/prof2_$TAG; rm -rf $d2
(cd /tmp && LNN_NO_WGRAD_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace -d $d2 -o r -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --other-workloads none > $OUT/prof2_bench.json 2> $OUT/prof2.err)
python tools/rocpd_stats.py $(find $d2 -name "*.db" | head -1) > $OUT/kernel_stats_serialized.txt 2>&1
timeout 300 python tools/layer_table.py --steps 6 > $OUT/layer_table.txt 2> $OUT/layer_table.err; tail -14 $OUT/layer_table.txt
