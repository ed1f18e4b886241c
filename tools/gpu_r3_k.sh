#!/bin/bash
# round 3, call K: what do the per-plane barrier and the partial-sum exchange cost in v9, and the per-tile barrier in wgrad v5?
# TIMING experiments with deliberately wrong kernels (tools/exp/*.so built with -DLNN_V9_EXPERIMENT_* / -DLNN_WG_EXPERIMENT_*).
TAG=${1:-r3k}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
LIB=lifelong-nnunet_amd/csrc/liblnn_hip.so
cp $LIB /tmp/liblnn_hip_orig.so
for v in orig nobar noex nobar_noex; do
  [ $v = orig ] && cp /tmp/liblnn_hip_orig.so $LIB || cp tools/exp/liblnn_hip_$v.so $LIB
  echo "== v9 variant: $v"
  timeout 120 python tools/kbench.py --layers enc0.1,dec4.0cat,enc1.1,enc2.1 --which fwd,dgrad --iters 10 2>&1 | grep -v amdgpu.ids | tail -4
done | tee $OUT/v9_barrier_exchange.txt
for v in orig wgnobar; do
  [ $v = orig ] && cp /tmp/liblnn_hip_orig.so $LIB || cp tools/exp/liblnn_hip_$v.so $LIB
  echo "== wgrad variant: $v"
  timeout 120 python tools/kbench.py --layers enc0.1,dec4.0,enc1.1,dec3.0 --which wgrad --iters 10 2>&1 | grep -v amdgpu.ids | tail -4
done | tee $OUT/wgrad_barrier.txt
cp /tmp/liblnn_hip_orig.so $LIB
