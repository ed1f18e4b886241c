#!/bin/bash
OUT=$PWD/gpurun_out/g; mkdir -p $OUT; export TMPDIR=/tmp
for dp in 0 1; do
  d=/tmp/prof_dp$dp; rm -rf $d
  (cd /tmp && LNN_FORCE_DP=$dp timeout 600 rocprofv3 --kernel-trace -d $d -o r -- python $OLDPWD/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/bench_dp$dp.json 2> $OUT/prof_dp$dp.err)
  python tools/rocpd_stats.py $(find $d -name "*.db" | head -1) > $OUT/stats_dp$dp.txt 2>&1
  head -3 $OUT/stats_dp$dp.txt; grep -i "rccl\|nccl\|unpack\|wgrad_s1\|AllReduce\|copy" $OUT/stats_dp$dp.txt | cut -c1-160
done
python - <<'PY'
import time, os, torch, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
import torch.distributed as dist
from lifelong_nnunet_amd import get_trainer_class
import bench
for dp in (0,1):
    os.environ["LNN_FORCE_DP"]=str(dp)
    if dp and not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    plans=dict(bench.WORKLOADS["c2"][0])
    def provider(task, split, p):
        from lifelong_nnunet_amd.training.network_training.multihead.nnUNetTrainerMultiHead import default_data_provider
        return bench.ResidentBatches(default_data_provider(task, split, p), torch.device("cuda:0"))
    tr=get_trainer_class("sequential")("seg_outputs","A",plans=plans,data_provider=provider,device="cuda:0")
    tr.initialize(True, num_epochs=1000); tr.network.train()
    for _ in range(3): tr.run_iteration(tr.tr_gen, True)
    torch.cuda.synchronize()
    # host enqueue time: run the iteration without the final fetch
    t0=time.perf_counter(); n=5
    for _ in range(n):
        l=tr.run_iteration(tr.tr_gen, True, False, False)   # detach=False: no host sync
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print(f"FORCE_DP={dp}: host enqueue {1e3*(t1-t0)/n:.2f} ms/step, total {1e3*(t2-t0)/n:.2f} ms/step")
    del tr
PY
