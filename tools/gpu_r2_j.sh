timeout 900 python -m pytest tests/test_training_gpu.py tests/test_cl_gpu.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused seg', d['ms_per_step'], d['value'])"
LNN_NO_FUSED_SEG=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('separate', d['ms_per_step'], d['value'])"
done
