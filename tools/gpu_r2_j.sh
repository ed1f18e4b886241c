mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "RCCL\|NCCL" > gpurun_out/full_gpu.log; tail -4 gpurun_out/full_gpu.log
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('v4', d['ms_per_step'], d['value'])"
LNN_WGRAD_V2=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('v2', d['ms_per_step'], d['value'])"
done
