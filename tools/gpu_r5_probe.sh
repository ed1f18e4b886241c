#!/bin/bash
# round-5 probe lease: v7 tile kernel with / without the padding-plane skip (isolated launches and the step, alternating)
TAG=${1:-r5probe2}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -q -m gpu -k "conv or trilinear or split or gen" --timeout=300 > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for r in 0 1 0 1; do
  echo "== LNN_V7_NO_ZSKIP=$r" >> $OUT/kbench.txt
  LNN_V7_NO_ZSKIP=$r timeout 120 python tools/kbench.py --layers enc3.1,dec1.0,enc5.1 --which fwd,dgrad --iters 20 2>&1 | grep -v amdgpu >> $OUT/kbench.txt
done
cat $OUT/kbench.txt | cut -c1-170
for n in skip_a noskip_a skip_b noskip_b; do
  e="X=1"; case $n in noskip*) e="LNN_V7_NO_ZSKIP=1";; esac
  env $e timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras > $OUT/bench_$n.json 2> $OUT/bench_$n.err
  python -c "import json; d=json.load(open('$OUT/bench_$n.json')); print('$n', round(d['ms_per_step'],3), round(d['value'],2))"
done
