#!/bin/bash
# round 3, call J: library-GEMM calibration of the practical fp16 MFMA roof (plain + PMC), and the PMC passes over bench.py again
# with the duration-clustered listing (several layers share one launch grid since the XCD-aware weight-gradient order)
TAG=${1:-r3j}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 200 python tools/gemm_roof.py 2>&1 | grep -v amdgpu.ids | tee $OUT/gemm_roof.txt
d=/tmp/pmc_gemm_$TAG; rm -rf $d
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d $d -o r -- python $OLDPWD/tools/gemm_roof.py > $OUT/pmc_gemm.log 2>&1)
python tools/rocpd_pmc.py $(find $d -name "*.db" | head -1) --by-grid > $OUT/pmc_gemm.txt 2>&1; head -24 $OUT/pmc_gemm.txt | cut -c1-150
run() {  # name, counters...
  name=$1; shift
  d=/tmp/pmc_${name}_$TAG; rm -rf $d
  (cd /tmp && LNN_NO_WGRAD_OVERLAP=1 timeout 400 rocprofv3 --kernel-trace --pmc "$@" -d $d -o r -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --other-workloads none > $OUT/pmc_$name.log 2>&1)
  python tools/rocpd_pmc.py $(find $d -name "*.db" | head -1) --by-grid > $OUT/pmc_$name.txt 2>&1
  echo "pass $name: $(grep -c '^==' $OUT/pmc_$name.txt) kernel groups"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run mfma GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES
python tools/pmc_traffic.py $OUT $OUT/pmc_traffic.json 2>&1 | head -5
python tools/pmc_mfma_clock.py $OUT/pmc_mfma.txt $OUT/pmc_fetch.txt $OUT/pmc_mfma_clock.json 2>&1 | head -24
