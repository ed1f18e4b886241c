#!/bin/bash
# round 3: PMC passes over the bench step itself (every kernel of the step, weight gradients on the main stream), separate
# rocprofv3 runs per counter set as MI355X_MICROARCH.md prescribes (--kernel-trace only next to --pmc)
TAG=${1:-r3pmc}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
run() {  # name, counters...
  name=$1; shift
  d=/tmp/pmc_$name; rm -rf $d
  (cd /tmp && LNN_NO_WGRAD_OVERLAP=1 timeout 400 rocprofv3 --kernel-trace --pmc "$@" -d $d -o r -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --other-workloads none > $OUT/pmc_$name.log 2>&1)
  db=$(find $d -name "*.db" | head -1)
  python tools/rocpd_pmc.py $db --by-grid > $OUT/pmc_$name.txt 2>&1
  echo "pass $name: $(grep -c '^==' $OUT/pmc_$name.txt) kernel groups"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
head -30 $OUT/pmc_fetch.txt
