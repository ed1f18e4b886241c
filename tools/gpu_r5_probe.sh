#!/bin/bash
# round-5 probe lease: v7 tile kernel with half steps (new library) vs padding-plane skip only (liblnn_hip_base.so), alternating
TAG=${1:-r5probe3}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
BASE=$PWD/lifelong-nnunet_amd/csrc/liblnn_hip_base.so
timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -q -m gpu -k "conv or trilinear or split or gen or variant" --timeout=300 > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log; grep -E "^FAILED" $OUT/pytest.log | head
for r in new base new base; do
  echo "== $r" >> $OUT/kbench.txt
  e="X=1"; [ $r = base ] && e="LNN_LIB_PATH=$BASE"
  env $e timeout 120 python tools/kbench.py --layers enc3.1,dec1.0,dec2.0 --which fwd,dgrad --iters 20 2>&1 | grep -v amdgpu >> $OUT/kbench.txt
done
cat $OUT/kbench.txt | cut -c1-170
for n in new_a base_a new_b base_b; do
  e="X=1"; case $n in base*) e="LNN_LIB_PATH=$BASE";; esac
  env $e timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras > $OUT/bench_$n.json 2> $OUT/bench_$n.err
  python -c "import json; d=json.load(open('$OUT/bench_$n.json')); print('$n', round(d['ms_per_step'],3), round(d['value'],2))"
done
