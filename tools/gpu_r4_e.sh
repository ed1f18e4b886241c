#!/bin/bash
TAG=${1:-r4e}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_plans_gpu.py tests/test_kernels_gpu.py tests/test_training_gpu.py -q -s --timeout=300 > $OUT/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed|Error" $OUT/pytest.log | tail -12; grep -E "3-step update|worst tensors" $OUT/pytest.log | tail -20
timeout 120 python tools/microbench_reductions.py 2>/dev/null | tail -1 | tee $OUT/microbench.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --other-workloads none > $OUT/bench.json 2> $OUT/bench.err
python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'],d['ms_per_step'])"
LNN_GEN=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --other-workloads none > $OUT/bench_nogen.json 2> $OUT/bench_nogen.err
python -c "import json;d=json.load(open('$OUT/bench_nogen.json'));print('LNN_GEN=0',d['value'],d['ms_per_step'])"
