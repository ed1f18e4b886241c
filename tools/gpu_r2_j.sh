timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_training_gpu.py tests/test_dp_gpu.py -x -q 2>&1 | grep -v "RCCL\|NCCL" | tail -3
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new', d['ms_per_step'], d['value'])"
LNN_PACK_ELEMENTWISE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old pack/unpack', d['ms_per_step'], d['value'])"
