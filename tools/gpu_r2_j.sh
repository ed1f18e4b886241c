timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_training_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('splitk', d['ms_per_step'], d['value'])"
LNN_NO_SPLITK=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no splitk', d['ms_per_step'], d['value'])"
done
