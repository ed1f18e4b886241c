#!/bin/bash
# fused normalisation-backward reduce in the v9 data gradient: parity, isolated A/B, whole-step A/B
OUT=$PWD/gpurun_out/${1:-r4red}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -q -k "fused_norm_backward_reduce or fused_reduce" 2>&1 | tail -25 > $OUT/pytest_red.log
tail -8 $OUT/pytest_red.log
timeout 300 python tools/kbench.py --layers enc0.1,enc1.1 --which dgrad,dgrad_red,dgrad_sums,in_bwd,in_apply --iters 10 > $OUT/kbench_red.txt 2>&1
cat $OUT/kbench_red.txt
for nf in 1 0; do
LNN_NO_FUSED_IN_BWD_REDUCE=$nf timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --other-workloads none > $OUT/bench_nofuse$nf.json 2> $OUT/bench_nofuse$nf.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_nofuse$nf.json").read().strip().splitlines()[-1]); print("no_fuse=$nf", d["value"], d["ms_per_step"])
PY
done
