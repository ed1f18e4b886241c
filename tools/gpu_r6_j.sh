#!/bin/bash
# round 6, call J: [1,3,3] weight gradients as the centre slice of the 3x3x3 weight gradient (parity, prostate-shaped plan A/B)
TAG=${1:-r6j}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gen_gpu.py tests/test_plans_gpu.py -q -m gpu --timeout=600 > $OUT/pytest_gen.log 2>&1; tail -6 $OUT/pytest_gen.log
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu --timeout=600 -k "first_layer" > $OUT/pytest_first.log 2>&1; tail -2 $OUT/pytest_first.log
P="--workload prostate --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras --other-workloads none"
for rep in 1 2; do
  LNN_K133_WGRAD_333=0 timeout 300 python bench.py $P > $OUT/bench_pro_gen_$rep.json 2> $OUT/bench_pro_gen_$rep.err
  python -c "import json;d=json.load(open('$OUT/bench_pro_gen_$rep.json'));print('prostate LNN_K133_WGRAD_333=0 rep $rep', round(d['ms_per_step'],3))"
  timeout 300 python bench.py $P > $OUT/bench_pro_333_$rep.json 2> $OUT/bench_pro_333_$rep.err
  python -c "import json;d=json.load(open('$OUT/bench_pro_333_$rep.json'));print('prostate default rep $rep', round(d['ms_per_step'],3))"
done
timeout 300 python bench.py --workload prostate --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --other-workloads none > $OUT/bench_pro_parity.json 2> $OUT/bench_pro_parity.err
python -c "import json;d=json.load(open('$OUT/bench_pro_parity.json'));print('prostate parity', round(d['ms_per_step'],3), json.dumps(d.get('parity',{}))[:600])"
timeout 300 python tools/layer_table.py --steps 6 --workload prostate > $OUT/layer_table_prostate.txt 2> $OUT/layer_table_prostate.err; head -24 $OUT/layer_table_prostate.txt | cut -c1-120
