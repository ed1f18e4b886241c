#!/bin/bash
# round 3, call R: per-wave phase split of the stride-1 weight gradient (LNN_WGRAD_DEBUG=4), DMA issue step 1 vs 2
TAG=${1:-r3r}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
for st in 2 1; do echo "== LNN_WGRAD_V4_STEP=$st"; LNN_WGRAD_V4_STEP=$st timeout 120 python tools/kbench.py --layers dec4.0,enc0.1,enc1.1 --which wgrad --iters 20 --wgrad-phases 2>&1 | grep -v amdgpu.ids | tail -6; done | tee $OUT/wgrad_phases.txt
