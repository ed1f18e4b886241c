#!/bin/bash
# round 6, call A: parity of the macro-tile kernel (both variants), per-layer timings against the shipped selection, power-ceiling arms
TAG=${1:-r6a}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
K="deep_layers or macro_tile or every_stride1 or cat_ops or splitk_small"
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "$K" --timeout=600 > $OUT/pytest_mt_pipe1.log 2>&1; tail -5 $OUT/pytest_mt_pipe1.log
LNN_MT_PIPE=0 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "deep_layers or macro_tile or every_stride1" --timeout=600 > $OUT/pytest_mt_pipe0.log 2>&1; tail -3 $OUT/pytest_mt_pipe0.log
L="enc3.1,dec1.0cat,enc4.1,dec0.0cat,enc5.1,enc2.1,dec2.0cat"
for rep in 1 2; do
echo "== default selection (rep $rep)"; timeout 300 python tools/kbench.py --layers $L --which fwd_st,dgrad --iters 20 2>&1 | tail -8
echo "== macro-tile, PIPE (rep $rep)"; timeout 300 python tools/kbench.py --layers $L --which fwd_st,dgrad --iters 20 --force 10 2>&1 | tail -8
echo "== macro-tile, no PIPE (rep $rep)"; LNN_MT_PIPE=0 timeout 300 python tools/kbench.py --layers $L --which fwd_st,dgrad --iters 20 --force 10 2>&1 | tail -8
done > $OUT/kbench_mt.txt 2>&1
cat $OUT/kbench_mt.txt
timeout 300 python tools/power_ceiling.py --seconds 1.5 --json $OUT/power_ceiling.json > $OUT/power_ceiling.txt 2>&1; cat $OUT/power_ceiling.txt
for fill in zero random; do
  dd=/tmp/pmc_pc_$fill; rm -rf $dd
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $dd -o r -- python $OLDPWD/tools/power_ceiling.py --seconds 0.4 --fill $fill > $OUT/pmc_pc_$fill.log 2>&1)
  python tools/rocpd_pmc.py $(find $dd -name "*.db" | head -1) --by-grid > $OUT/pmc_pc_$fill.txt 2>&1
  grep -A4 "igemm_conv_s1_v9\|igemm_wgrad_s1_v5\|Cijk\|gemm" $OUT/pmc_pc_$fill.txt | head -40
done
