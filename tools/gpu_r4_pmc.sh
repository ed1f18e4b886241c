#!/bin/bash
# round-4 evidence, part 2: PMC passes over the bench step itself (every kernel of the step, weight gradients on the main stream),
# separate rocprofv3 runs per counter set as MI355X_MICROARCH.md prescribes (--kernel-trace only next to --pmc), then the two JSON
# files bench.py quotes (stamped with the sha256 of the liblnn_hip.so they were collected with), and the library-GEMM calibration
TAG=${1:-r4pmc}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
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
run mfma GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES
python tools/pmc_traffic.py $OUT $OUT/pmc_traffic.json 2>&1 | head -6
python tools/pmc_mfma_clock.py $OUT/pmc_mfma.txt $OUT/pmc_fetch.txt $OUT/pmc_mfma_clock.json 2>&1 | head -16
timeout 200 python tools/gemm_roof.py > $OUT/library_gemm_roof.txt 2>&1; tail -4 $OUT/library_gemm_roof.txt
