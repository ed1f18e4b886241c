#!/bin/bash
# PMC passes over tools/kbench.py for the v9 conv kernel (separate rocprofv3 runs, --kernel-trace only, as the guide prescribes)
# usage: bash tools/gpu_r2_pmc.sh <tag> <layers> <which>
TAG=${1:-b}; LAYERS=${2:-dec4.0cat,enc0.1}; WHICH=${3:-fwd}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
run() {  # name, counters...
  name=$1; shift
  d=/tmp/pmc_$name; rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $d -o r -- python $OLDPWD/tools/kbench.py --layers $LAYERS --which $WHICH --iters 3 > $OUT/pmc_$name.log 2>&1)
  db=$(find $d -name "*.db" | head -1)
  python tools/rocpd_pmc.py $db > $OUT/pmc_$name.txt 2>&1
  python tools/rocpd_stats.py $db igemm >> $OUT/pmc_$name.txt 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run sq2 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
grep -h -A12 "v9_kernel" $OUT/pmc_sq1.txt | head -40
grep -h -A12 "v9_kernel" $OUT/pmc_sq2.txt | head -40
grep -h -A2 "v9_kernel" $OUT/pmc_fetch.txt $OUT/pmc_write.txt | head
