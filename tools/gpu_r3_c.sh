#!/bin/bash
# round 3, call C: new kernels (v9 <8,1,1> for 128 channels, z-streaming stride-2 conv, fused seg backward) -- tests + single-layer timings
TAG=${1:-r3c}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > $OUT/pytest_kernels.log 2>&1; tail -4 $OUT/pytest_kernels.log
echo "== C=128 layers: v9 <8,1,1>"
timeout 200 python tools/kbench.py --layers dec3.0,enc2.1,dec2.0d --which fwd,dgrad --iters 10 2>&1 | tail -4
echo "== same layers, v9 forbidden for C=128 (v8/v7)"
LNN_CONV_V9=0 timeout 200 python tools/kbench.py --layers dec3.0,enc2.1,dec2.0d --which fwd,dgrad --iters 10 2>&1 | tail -4
echo "== stride-2 fwd: streaming vs tile"
timeout 200 python tools/kbench.py --layers enc1.0s2,enc2.0s2 --which fwd --iters 10 --down2 1 2>&1 | tail -3
timeout 200 python tools/kbench.py --layers enc1.0s2,enc2.0s2 --which fwd --iters 10 --down2 0 2>&1 | tail -3
timeout 600 python -m pytest tests/test_training_gpu.py tests/test_fullsize_gpu.py tests/test_trainer_goldens_gpu.py -x -q -m gpu > $OUT/pytest_train.log 2>&1; tail -4 $OUT/pytest_train.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --other-workloads none > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err | cut -c1-300
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d["value"], d["ms_per_step"])
print({k:(round(v["launch_ms_in_step"],3), round(v["launch_ms_isolated"],3), round(v["launch_ms_in_timed_steps_two_streams"],3)) for k,v in d["roofline"]["families"].items()})
PY
d2=/tmp/prof2_$TAG; rm -rf $d2
(cd /tmp && LNN_NO_WGRAD_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace -d $d2 -o r -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/prof2_bench.json 2> $OUT/prof2.err)
python tools/rocpd_stats.py $(find $d2 -name "*.db" | head -1) > $OUT/kernel_stats_serialized.txt 2>&1
head -32 $OUT/kernel_stats_serialized.txt | cut -c1-170
