#!/bin/bash
# round 3, call A: kernel tests of the new fused entries, training parity, bench line, serialised kernel table
TAG=${1:-r3a}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_training_gpu.py tests/test_trainer_goldens_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err | cut -c1-300; cut -c1-400 $OUT/bench.json
LNN_NO_FUSED_SEG_BWD=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/bench_nofuse.json 2> $OUT/bench_nofuse.err; cut -c1-200 $OUT/bench_nofuse.json
d2=/tmp/prof2_$TAG; rm -rf $d2
(cd /tmp && LNN_NO_WGRAD_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace -d $d2 -o r -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/prof2_bench.json 2> $OUT/prof2.err)
python tools/rocpd_stats.py $(find $d2 -name "*.db" | head -1) > $OUT/kernel_stats_serialized.txt 2>&1
head -40 $OUT/kernel_stats_serialized.txt | cut -c1-170
