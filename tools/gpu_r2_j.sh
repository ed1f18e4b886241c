timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "wgrad" 2>&1 | tail -2
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new', d['ms_per_step'], d['value'])"
LNN_WGRAD_NOSHARE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('noshare', d['ms_per_step'], d['value'])"
done
