timeout 600 python -m pytest tests/test_dp_gpu.py -x -q -s 2>&1 | tail -25
for dp in 0 1; do LNN_FORCE_DP=$dp timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('FORCE_DP=$dp', d['ms_per_step'], d['value'])"; done
for dp in 0 1; do LNN_FORCE_DP=$dp timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('FORCE_DP=$dp', d['ms_per_step'], d['value'])"; done
