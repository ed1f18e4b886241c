#!/bin/bash
# round 6, call I: [1,3,3] convolutions on the z-streaming kernel with permuted axes (parity, prostate-shaped plan A/B, layer table)
TAG=${1:-r6i}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 600 python -m pytest tests/test_gen_gpu.py -q -m gpu -x --timeout=300 -k "k133" > $OUT/pytest_k133.log 2>&1; tail -12 $OUT/pytest_k133.log
timeout 900 python -m pytest tests/test_gen_gpu.py tests/test_plans_gpu.py tests/test_kernels_gpu.py -q -m gpu --timeout=600 > $OUT/pytest_gen.log 2>&1; tail -4 $OUT/pytest_gen.log
P="--workload prostate --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras --other-workloads none"
for rep in 1 2; do
  LNN_CONV_K133_V9=0 timeout 300 python bench.py $P > $OUT/bench_pro_gen_$rep.json 2> $OUT/bench_pro_gen_$rep.err
  python -c "import json;d=json.load(open('$OUT/bench_pro_gen_$rep.json'));print('prostate LNN_CONV_K133_V9=0 rep $rep', round(d['ms_per_step'],3), json.dumps(d.get('parity',{}))[:300])"
  timeout 300 python bench.py $P > $OUT/bench_pro_v9_$rep.json 2> $OUT/bench_pro_v9_$rep.err
  python -c "import json;d=json.load(open('$OUT/bench_pro_v9_$rep.json'));print('prostate default rep $rep', round(d['ms_per_step'],3), json.dumps(d.get('parity',{}))[:300])"
done
timeout 300 python tools/layer_table.py --steps 6 --workload prostate > $OUT/layer_table_prostate.txt 2> $OUT/layer_table_prostate.err; head -30 $OUT/layer_table_prostate.txt | cut -c1-120; tail -26 $OUT/layer_table_prostate.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras --other-workloads none > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python -c "import json;d=json.load(open('$OUT/bench_c2.json'));print('c2', round(d['ms_per_step'],3))"
