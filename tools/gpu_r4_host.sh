#!/bin/bash
OUT=$PWD/gpurun_out/${1:-r4host}; mkdir -p $OUT
LNN_SAMPLE_LANES=0 timeout 300 python tools/host_overhead.py 2>/dev/null | tail -1 | tee -a $OUT/host.txt
LNN_SAMPLE_LANES=1 timeout 300 python tools/host_overhead.py 2>/dev/null | tail -1 | tee -a $OUT/host.txt
for b in 192 240; do
LNN_SAMPLE_LANES=1 LNN_LANE_CUS=$b timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --other-workloads none > $OUT/bench_$b.json 2>/dev/null
python -c "import json;d=json.load(open('$OUT/bench_$b.json'));print('lanes $b',round(d['value'],2),round(d['ms_per_step'],3))" | tee -a $OUT/host.txt
done
