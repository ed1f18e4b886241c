mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_plop_gpu.py -q 2>&1 | grep -v "RCCL\|NCCL" | tail -40 > gpurun_out/plop.log
tail -25 gpurun_out/plop.log
