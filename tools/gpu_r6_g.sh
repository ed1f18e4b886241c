#!/bin/bash
# round 6, call G / H: one-launch small-volume normalisation (parity, isolated timing, step A/B, kernel trace), column bands in the
# trace), column bands in the macro-tile kernel + z-streaming kernel from 16 planes (parity, prostate-shaped plan)
TAG=${1:-r6g}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 600 python -m pytest tests/test_small_volume_gpu.py -q -m gpu -x --timeout=300 > $OUT/pytest_small.log 2>&1; tail -3 $OUT/pytest_small.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x --timeout=600 -k "macro_tile or deep_layers or every_stride1 or fwd_in_stats or cat_ops or z_segments" > $OUT/pytest_mt.log 2>&1; tail -3 $OUT/pytest_mt.log
timeout 900 python -m pytest tests/test_plans_gpu.py tests/test_training_gpu.py tests/test_gen_gpu.py -q -m gpu -x --timeout=600 > $OUT/pytest_plans.log 2>&1; tail -3 $OUT/pytest_plans.log
LNN_IN_SMALL=0 timeout 200 python tools/kbench_small.py 2>&1 | tee $OUT/kbench_small_multi.txt
timeout 200 python tools/kbench_small.py 2>&1 | tee $OUT/kbench_small_one.txt
B="--steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extras --other-workloads none"
for rep in 1 2; do
  for sm in 0 1; do
    LNN_IN_SMALL=$sm timeout 300 python bench.py $B > $OUT/bench_small${sm}_$rep.json 2> $OUT/bench_small${sm}_$rep.err
    python -c "import json;d=json.load(open('$OUT/bench_small${sm}_$rep.json'));print('LNN_IN_SMALL=$sm rep $rep', round(d['ms_per_step'],3))"
  done
done
P="--workload prostate --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras --other-workloads none"
for rep in 1 2; do
  timeout 300 python bench.py $P > $OUT/bench_pro_$rep.json 2> $OUT/bench_pro_$rep.err
  python -c "import json;d=json.load(open('$OUT/bench_pro_$rep.json'));print('prostate rep $rep', round(d['ms_per_step'],3), json.dumps(d.get('parity',{}))[:300])"
done
timeout 300 python tools/layer_table.py --steps 6 --workload prostate > $OUT/layer_table_prostate.txt 2> $OUT/layer_table_prostate.err; head -14 $OUT/layer_table_prostate.txt | cut -c1-120
d=/tmp/prof_$TAG; rm -rf $d
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $d -o r -- python $OLDPWD/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-extras --other-workloads none > $OUT/prof_bench.json 2> $OUT/prof.err)
db=$(find $d -name "*.db" | head -1)
python tools/rocpd_stats.py $db > $OUT/kernel_stats.txt 2>&1; grep -E "in_small|splitk_finalize|in_stats|in_lrelu_fwd|in_lrelu_bwd" $OUT/kernel_stats.txt | cut -c1-200
python tools/step_timeline.py $db --step 5 > $OUT/timeline.txt 2>&1; tail -4 $OUT/timeline.txt
