#!/bin/bash
# round-4 lease B: generic flattened-voxel kernels -- parity tests, per-layer A/B against the tile kernels, step A/B
TAG=${1:-r4b}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 600 python -m pytest tests/test_gen_gpu.py -q --timeout=300 -x > $OUT/pytest_gen.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed|Error" $OUT/pytest_gen.log | tail -12
SMALL=enc4.1,dec0.0,enc5.1,enc4.0s2,enc5.0s2,enc3.1,dec1.0,enc3.0s2
for g in 0 1; do
  echo "== gen $g" >> $OUT/kbench.txt
  timeout 300 python tools/kbench.py --gen $g --layers $SMALL,up0,up1,up2 --iters 20 >> $OUT/kbench.txt 2>&1
done
cat $OUT/kbench.txt | cut -c1-200
for mv in 0 4096 20000; do
  LNN_GEN_MAXVOX=$mv timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --other-workloads none > $OUT/bench_mv$mv.json 2> $OUT/bench_mv$mv.err
  python -c "import json;d=json.load(open('$OUT/bench_mv$mv.json'));print('maxvox',$mv,d['value'],d['ms_per_step'],d['config']['loss'])"
done
timeout 900 python -m pytest tests -q -m gpu --timeout=300 -x --deselect tests/test_gen_gpu.py > $OUT/pytest_gpu.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest_gpu.log | tail -8
