#!/bin/bash
# round-5 probe lease: conditioning of the fused normalisation-backward reduce; run-to-run spread of the C2 step (three 20-step runs)
TAG=${1:-r5probe}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python tools/probes/fused_reduce_conditioning.py > $OUT/fused_reduce_conditioning.txt 2>&1; cat $OUT/fused_reduce_conditioning.txt
for i in 1 2 3; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/bench_$i.json 2> $OUT/bench_$i.err
  python -c "import json; d=json.load(open('$OUT/bench_$i.json')); print('run $i', round(d['ms_per_step'],3), round(d['value'],2), round(d['ms_per_step_h2d_inclusive'],3), round(d['config']['eager_loss_fetch']['ms_per_step'],3))"
done
