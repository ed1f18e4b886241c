#!/bin/bash
# round-5 lease C: the tests that failed in lease B + the new ones, A/B of the lazy-normalisation experiment, the full bench line
# (process-wide side streams: c3 / c4 / c5 no longer slower than c2?), CPU thread sweep
TAG=${1:-r5c}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 1200 python -m pytest tests/test_cl_gpu.py tests/test_dp2_gpu.py tests/test_dp_gpu.py tests/test_training_gpu.py tests/test_kernels_gpu.py tests/test_trainer_goldens_gpu.py -q -m gpu --timeout=600 > $OUT/pytest_gpu.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest_gpu.log | tail -15
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
b lazy_in LNN_EXP_LAZY_IN=1
b default1 X=1
b lazy_in1 LNN_EXP_LAZY_IN=1
b no_c1_stream LNN_NO_C1_WGRAD_STREAM=1
b default2 X=1
timeout 900 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; tail -2 $OUT/bench_full.err | cut -c1-300
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_full.json"))
    print(d["value"], d["ms_per_step"], d["roofline"]["slowest_family"], round(d["roofline"]["frac"],3), d["config"]["conv_stack_frac_of_mfma_peak"], d["config"].get("eager_loss_fetch"), d["config"].get("h2d_inclusive"))
    for k,v in d.get("other_workloads",{}).items():
        print(k, v.get("value"), v.get("ms_per_step"), v.get("same_batch_predictions_patches_per_s"), v.get("error"), json.dumps((v.get("parity") or {}).get("gates")), json.dumps(((v.get("parity") or {}).get("full_iteration") or {}).get("gates")))
except Exception as e: print("ERR", e)
PY
timeout 400 python tools/cpu_thread_sweep.py > $OUT/cpu_thread_sweep.txt 2>&1; cat $OUT/cpu_thread_sweep.txt
