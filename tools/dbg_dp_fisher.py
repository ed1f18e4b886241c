import os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from lifelong_nnunet_amd import get_trainer_class
from lifelong_nnunet_amd.synthetic import make_patch_batch
DEV = "cuda:0"
TOY = {"patch_size": (16, 32, 16), "batch_size": 2, "num_pool": 2, "base_num_features": 8, "num_classes": 3, "num_input_channels": 1, "synthetic_period": 4}
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
def trainer(force):
    os.environ["LNN_FORCE_DP"] = "1" if force else "0"
    tr = get_trainer_class("ewc")("seg_outputs", "taskA", plans=dict(TOY), device=DEV, batch_dice=True, fisher_mode="accumulate", deterministic_wgrad=(os.environ.get("DBG_DET", "0") == "1"))
    tr.initialize(True, num_epochs=1)
    os.environ["LNN_FORCE_DP"] = "0"
    tr.num_batches_per_epoch, tr.num_val_batches_per_epoch = 3, 0
    return tr
def batches(n, seed):
    out = []
    for i in range(n):
        d, t = make_patch_batch(2, TOY["patch_size"], 2, seed=seed + i)
        out.append({"data": d, "target": t, "keys": ["a", "b"]})
    return out
trs = [trainer(False), trainer(False), trainer(True), trainer(True)]
sd = trs[0].network.state_dict()
for t in trs[1:]:
    t.network.load_state_dict(sd); t.mh_network.update_after_iteration()
data = batches(6, 70)
for tr in trs:
    tr.data_provider = lambda task, split, plans: iter(data)
    tr.reinitialize("taskA"); tr.run_training("taskA")
names = [n for n in trs[0].fisher["taskA"] if trs[0].fisher["taskA"][n].numel() > 1]
def rel(x, y): return float((x - y).norm() / (y.norm() + 1e-30))
print("losses", [t.all_tr_losses for t in trs])
for i, j, lab in ((0, 1, "plain vs plain"), (2, 3, "dp vs dp"), (0, 2, "plain vs dp")):
    fa = torch.cat([trs[i].fisher["taskA"][n].reshape(-1) for n in names]); fb = torch.cat([trs[j].fisher["taskA"][n].reshape(-1) for n in names])
    print(lab, "fisher rel", rel(fa, fb), "theta rel", rel(trs[i].network.arena.theta, trs[j].network.arena.theta))
    worst = sorted(((rel(trs[i].fisher["taskA"][n], trs[j].fisher["taskA"][n]), n) for n in names), reverse=True)[:6]
    print("   worst:", [(round(a, 6), n) for a, n in worst])
dist.destroy_process_group()
