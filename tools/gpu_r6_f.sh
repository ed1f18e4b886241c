#!/bin/bash
# round 6, call F: one-launch normalisation of the small volumes (parity + step A/B), z-streaming kernel on 20-plane volumes (prostate plan A/B)
TAG=${1:-r6f}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 600 python -m pytest tests/test_small_volume_gpu.py -q -m gpu -x --timeout=300 > $OUT/pytest_small.log 2>&1; tail -15 $OUT/pytest_small.log
timeout 900 python -m pytest tests/test_training_gpu.py tests/test_trainer_goldens_gpu.py tests/test_dp_gpu.py -q -m gpu -x --timeout=600 > $OUT/pytest_train.log 2>&1; tail -5 $OUT/pytest_train.log
B="--steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extras --other-workloads none"
for rep in 1 2; do
  for sm in 0 1; do
    LNN_IN_SMALL=$sm timeout 300 python bench.py $B > $OUT/bench_small${sm}_$rep.json 2> $OUT/bench_small${sm}_$rep.err
    python -c "import json;d=json.load(open('$OUT/bench_small${sm}_$rep.json'));print('LNN_IN_SMALL=$sm rep $rep', round(d['ms_per_step'],3))"
  done
done
P="--workload prostate --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras --other-workloads none"
for rep in 1 2; do
  timeout 300 python bench.py $P > $OUT/bench_pro_default_$rep.json 2> $OUT/bench_pro_default_$rep.err
  python -c "import json;d=json.load(open('$OUT/bench_pro_default_$rep.json'));print('prostate default rep $rep', round(d['ms_per_step'],3), d.get('parity',{}))" 2>&1 | cut -c1-300
  LNN_CONV_V9=1 timeout 300 python bench.py $P > $OUT/bench_pro_v9_$rep.json 2> $OUT/bench_pro_v9_$rep.err
  python -c "import json;d=json.load(open('$OUT/bench_pro_v9_$rep.json'));print('prostate LNN_CONV_V9=1 rep $rep', round(d['ms_per_step'],3), d.get('parity',{}))" 2>&1 | cut -c1-300
done
LNN_CONV_V9=1 timeout 300 python tools/layer_table.py --steps 6 --workload prostate > $OUT/layer_table_prostate_v9.txt 2> $OUT/layer_table_prostate_v9.err; head -40 $OUT/layer_table_prostate_v9.txt
