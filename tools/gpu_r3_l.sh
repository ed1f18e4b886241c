#!/bin/bash
# round 3, call L: wider finalize / sums kernels of the InstanceNorm family: tests, step time, their share in a serialised trace
TAG=${1:-r3l}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_training_gpu.py -q -m gpu --timeout=120 > $OUT/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest.log | tail -8
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --other-workloads none 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], d['value'])"
d2=/tmp/prof2_$TAG; rm -rf $d2
(cd /tmp && LNN_NO_WGRAD_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace -d $d2 -o r -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --other-workloads none > $OUT/prof2_bench.json 2> $OUT/prof2.err)
python tools/rocpd_stats.py $(find $d2 -name "*.db" | head -1) > $OUT/kernel_stats_serialized.txt 2>&1
grep -E "sums_kernel|finalize|in_stats_kernel|gradnorm|pack|up2|c1_" $OUT/kernel_stats_serialized.txt | cut -c1-170
