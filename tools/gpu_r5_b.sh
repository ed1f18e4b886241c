#!/bin/bash
# round-5 lease B: full GPU suite on the new engine / kernels, A/B of the round-5 switches, step timeline, the full default bench line
TAG=${1:-r5b}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 1500 python -m pytest tests -q -m gpu --timeout=600 > $OUT/pytest_gpu.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest_gpu.log | tail -15
b() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$name.json")); e = d["config"].get("eager_loss_fetch", {})
    print("$name", round(d["ms_per_step"], 3), "ms", round(d["value"], 2), "patches/s | eager fetch", round(e.get("ms_per_step", 0), 3), "| loss", d["config"]["loss"])
except Exception as e:
    print("$name ERR", e)
PY
}
b default0 X=1
b no_c1_stream LNN_NO_C1_WGRAD_STREAM=1
b no_pack_overlap LNN_NO_PACK_OVERLAP=1
b no_lazy_z LNN_NO_LAZY_TOP_Z=1
b default1 X=1
d=/tmp/prof_$TAG; rm -rf $d
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $d -o r -- python $OLDPWD/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/prof.json 2> $OUT/prof.err)
db=$(find $d -name "*.db" | head -1)
python tools/step_timeline.py $db --step 5 > $OUT/timeline.txt 2>&1; tail -4 $OUT/timeline.txt
python tools/rocpd_stats.py $db > $OUT/kernel_stats.txt 2>&1
timeout 900 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; tail -2 $OUT/bench_full.err | cut -c1-300
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_full.json"))
    print(d["value"], d["ms_per_step"], d["roofline"]["slowest_family"], round(d["roofline"]["frac"],3), d["config"]["conv_stack_frac_of_mfma_peak"], d["config"].get("eager_loss_fetch"), d["config"].get("h2d_inclusive"))
    print(d.get("cpu_baseline")); print(d.get("parity"))
    for k,v in d.get("other_workloads",{}).items():
        print(k, v.get("value"), v.get("ms_per_step"), v.get("error"), json.dumps(v.get("parity"))[:700])
except Exception as e: print("ERR", e)
PY
