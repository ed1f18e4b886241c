#!/bin/bash
# round-2 evidence: full GPU suite, bench lines of all workloads, kernel-trace --stats of the bench command, serialised table
TAG=${1:-final}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
bash tools/gpu_r2_bench.sh $TAG
d=/tmp/prof_$TAG; rm -rf $d
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $d -o r -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err)
db=$(find $d -name "*.db" | head -1); python tools/rocpd_stats.py $db > $OUT/kernel_stats.txt 2>&1; head -12 $OUT/kernel_stats.txt
find $d -name "*stats*" | head; cp $(find $d -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats.csv 2>/dev/null
d2=/tmp/prof2_$TAG; rm -rf $d2
(cd /tmp && LNN_NO_WGRAD_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace -d $d2 -o r -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/prof2_bench.json 2> $OUT/prof2.err)
python tools/rocpd_stats.py $(find $d2 -name "*.db" | head -1) > $OUT/kernel_stats_serialized.txt 2>&1
tail -2 $OUT/prof_bench.json | cut -c1-600
