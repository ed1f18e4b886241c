timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "in_stats or v9 or stride1" 2>&1 | tail -5
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for f in 1 0; do LNN_NO_FUSED_IN_STATS=$f timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NO_FUSED_IN_STATS=$f', d['ms_per_step'], d['value'])"; done
for f in 1 0; do LNN_NO_FUSED_IN_STATS=$f timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NO_FUSED_IN_STATS=$f', d['ms_per_step'], d['value'])"; done
