#!/bin/bash
# round 6, call L: rocprofv3 kernel trace of the prostate-shaped plan (which kernel serves which share of its step)
TAG=${1:-r6l}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
d=/tmp/prof_$TAG; rm -rf $d
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $d -o r -- python $OLDPWD/bench.py --workload prostate --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras --other-workloads none > $OUT/prof_bench.json 2> $OUT/prof.err)
db=$(find $d -name "*.db" | head -1)
python tools/rocpd_stats.py $db > $OUT/kernel_stats_prostate.txt 2>&1; head -30 $OUT/kernel_stats_prostate.txt | cut -c1-190
