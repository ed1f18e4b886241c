#!/bin/bash
# full GPU suite + bench + kernel-trace summary.  usage: bash tools/gpu_r2_full.sh <tag> [extra bench args]
TAG=${1:-full}; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 "$@" > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json | cut -c1-1500
d=/tmp/prof_$TAG; rm -rf $d
(cd /tmp && LNN_NO_WGRAD_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace -d $d -o r -- python $OLDPWD/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/prof_bench.json 2> $OUT/prof.err)
python tools/rocpd_stats.py $(find $d -name "*.db" | head -1) > $OUT/kernel_stats_serialized.txt 2>&1; head -40 $OUT/kernel_stats_serialized.txt
