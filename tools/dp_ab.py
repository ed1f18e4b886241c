"""A/B of the forced data-parallel step against the plain step, one process per arm (usage: python tools/dp_ab.py plain|dp|dp_late)."""
import time, os, torch, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29534")
import torch.distributed as dist
arm = sys.argv[1]
torch.cuda.set_device(0)
if arm == "dp":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
if arm == "dp_lazy":          # no device_id: the communicator is created by the first collective, after the engine's allocations
    dist.init_process_group("nccl", rank=0, world_size=1)
if arm == "dp_prealloc":      # eager communicator, but the caching allocator already holds pre-init segments
    x = torch.empty(24 << 30, dtype=torch.uint8, device="cuda:0"); del x
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from lifelong_nnunet_amd import get_trainer_class
import bench
def make(dp):
    os.environ["LNN_FORCE_DP"]=str(dp)
    plans=dict(bench.WORKLOADS["c2"][0])
    def provider(task, split, p):
        from lifelong_nnunet_amd.training.network_training.multihead.nnUNetTrainerMultiHead import default_data_provider
        return bench.ResidentBatches(default_data_provider(task, split, p), torch.device("cuda:0"))
    tr=get_trainer_class("sequential")("seg_outputs","A",plans=plans,data_provider=provider,device="cuda:0")
    tr.initialize(True, num_epochs=1000); tr.network.train()
    for _ in range(5): tr.run_iteration(tr.tr_gen, True)
    return tr
def run(tr, detach, n=20):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): tr.run_iteration(tr.tr_gen, True, False, detach)
    torch.cuda.synchronize(); return 1e3*(time.perf_counter()-t0)/n
if arm == "dp_late":
    tmp = make(0); del tmp
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
tr = make(0 if arm == "plain" else 1)
print(arm, "detach=True %.2f %.2f  detach=False %.2f" % (run(tr,True), run(tr,True), run(tr,False)))
