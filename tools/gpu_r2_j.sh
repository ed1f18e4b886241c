for i in 1 2; do
LNN_SAMPLE_LANES=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lanes', d['ms_per_step'], d['value'], d['config']['loss'])"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain', d['ms_per_step'], d['value'], d['config']['loss'])"
done
