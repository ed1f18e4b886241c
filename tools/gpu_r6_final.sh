#!/bin/bash
# round-6 evidence on the FINAL binary: full GPU suite, smoke, the default bench line (CPU baseline + parity blocks + other workloads),
# rocprofv3 --kernel-trace --stats of the bench command (two streams) and serialised, the step timeline, the per-layer table, PMC
# passes over the bench step (separate rocprofv3 runs per counter set, --kernel-trace only next to --pmc), library-GEMM calibration
TAG=${1:-r6final}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
sha256sum lifelong-nnunet_amd/csrc/liblnn_hip.so | tee $OUT/so_sha256.txt
timeout 1500 python -m pytest tests -q -m gpu --timeout=600 > $OUT/pytest_gpu.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest_gpu.log | tail -8
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > $OUT/bench_c2.json 2> $OUT/bench.err; tail -2 $OUT/bench.err | cut -c1-300
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_c2.json"))
    r=d["roofline"]
    print(d["value"], d["ms_per_step"], d.get("ms_per_step_h2d_inclusive"), r["slowest_family"], round(r["frac"],3), d["config"]["conv_stack_frac_of_mfma_peak"], d["config"]["eager_loss_fetch"]["ms_per_step"])
    print("frac_kernel_avg", r.get("frac_kernel_avg"), "power_w", r.get("power_w_in_timed_steps"), "library_gemm_frac", r.get("library_gemm_frac"), {k:(round(v["tflops"]),v["launches"]) for k,v in (r.get("all_stride1_launches_by_family") or {}).items()})
    print({k:(round(v["launch_ms_in_step"],3), round(v["launch_ms_isolated"],3), round(v["launch_ms_in_timed_steps_two_streams"],3)) for k,v in d["roofline"]["families"].items()})
    print(d.get("cpu_baseline")); print(d.get("parity"))
    for k,v in d.get("other_workloads",{}).items():
        print(k, v.get("value"), v.get("ms_per_step"), v.get("same_batch_predictions_patches_per_s"), v.get("error"), json.dumps((v.get("parity") or {}).get("gates")), json.dumps(((v.get("parity") or {}).get("full_iteration") or {}).get("gates")))
    print({k:(round(v["GBps"]),round(v["ms"],3)) for k,v in d.get("regulariser_kernels",{}).get("kernels",{}).items() if "GBps" in v})
except Exception as e: print("ERR", e)
PY
d=/tmp/prof_$TAG; rm -rf $d
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $d -o r -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --other-workloads none > $OUT/prof_bench.json 2> $OUT/prof.err)
db=$(find $d -name "*.db" | head -1)
python tools/rocpd_stats.py $db > $OUT/kernel_stats.txt 2>&1; head -8 $OUT/kernel_stats.txt | cut -c1-170
python tools/step_timeline.py $db --step 8 > $OUT/timeline.txt 2>&1; tail -4 $OUT/timeline.txt
cp $(find $d -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats.csv 2>/dev/null
d2=/tmp/prof2_$TAG; rm -rf $d2
(cd /tmp && LNN_NO_WGRAD_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace -d $d2 -o r -- python $OLDPWD/bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-roofline --no-extras --other-workloads none > $OUT/prof2_bench.json 2> $OUT/prof2.err)
python tools/rocpd_stats.py $(find $d2 -name "*.db" | head -1) > $OUT/kernel_stats_serialized.txt 2>&1; head -3 $OUT/kernel_stats_serialized.txt | cut -c1-170
timeout 300 python tools/layer_table.py --steps 6 > $OUT/layer_table.txt 2> $OUT/layer_table.err; tail -22 $OUT/layer_table.txt
run() {  # name, counters...
  name=$1; shift
  dd=/tmp/pmc_$name; rm -rf $dd
  (cd /tmp && LNN_NO_WGRAD_OVERLAP=1 timeout 400 rocprofv3 --kernel-trace --pmc "$@" -d $dd -o r -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --other-workloads none > $OUT/pmc_$name.log 2>&1)
  python tools/rocpd_pmc.py $(find $dd -name "*.db" | head -1) --by-grid > $OUT/pmc_$name.txt 2>&1
  echo "pass $name: $(grep -c '^==' $OUT/pmc_$name.txt) kernel groups"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run mfma GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES
python tools/pmc_traffic.py $OUT $OUT/pmc_traffic.json 2>&1 | head -6
python tools/pmc_mfma_clock.py $OUT/pmc_mfma.txt $OUT/pmc_fetch.txt $OUT/pmc_mfma_clock.json 2>&1 | head -16
timeout 200 python tools/gemm_roof.py > $OUT/library_gemm_roof.txt 2>&1; tail -4 $OUT/library_gemm_roof.txt
timeout 300 python tools/layer_table.py --steps 6 --workload prostate > $OUT/layer_table_prostate.txt 2> $OUT/layer_table_prostate.err; tail -8 $OUT/layer_table_prostate.txt
# the z-streaming kernel on 20-plane volumes of a Hippocampus-sized plan (C1: 40x56x40 patches): automatic rule 16 planes vs the old 32
for mp in 32 16 32 16; do
  LNN_CONV_V9_MIN_PLANES=$mp timeout 200 python bench.py --workload c1 --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-extras --other-workloads none > $OUT/bench_c1_minplanes$mp.json 2> $OUT/bench_c1.err
  python -c "import json;d=json.load(open('$OUT/bench_c1_minplanes$mp.json'));print('c1 LNN_CONV_V9_MIN_PLANES=$mp', round(d['ms_per_step'],3))"
done
