#!/bin/bash
# round-5 lease D: weight-gradient kernel A/B (running plane descriptors vs the descriptor arithmetic of rounds 2-4), isolated launches
# and in the step, alternating; a dispatch-rule check for the 64-output-channel tile kernel on the deep levels
TAG=${1:-r5d}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -q -m gpu -k "wgrad or trilinear" --timeout=300 > $OUT/pytest_wgrad.log 2>&1; tail -2 $OUT/pytest_wgrad.log
for r in 1 0 1 0; do
  echo "== LNN_WGRAD_RUN=$r" >> $OUT/kbench_wgrad.txt
  LNN_WGRAD_RUN=$r timeout 120 python tools/kbench.py --layers dec4.0,enc0.1,dec3.0,enc1.1,dec2.0,enc2.1,enc3.1,dec1.0,enc4.1 --which wgrad --iters 10 >> $OUT/kbench_wgrad.txt 2>&1
done
cat $OUT/kbench_wgrad.txt | cut -c1-160
b() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$name.json")); e = d["config"].get("eager_loss_fetch", {})
    print("$name", round(d["ms_per_step"], 3), "ms", round(d["value"], 2), "patches/s | eager fetch", round(e.get("ms_per_step", 0), 3), "| h2d", round(d.get("ms_per_step_h2d_inclusive", 0), 3))
except Exception as e:
    print("$name ERR", e)
PY
}
b run1_a LNN_WGRAD_RUN=1
b run0_a LNN_WGRAD_RUN=0
b run1_b LNN_WGRAD_RUN=1
b run0_b LNN_WGRAD_RUN=0
b v8_everywhere LNN_CONV_V8=1
b run1_c LNN_WGRAD_RUN=1
