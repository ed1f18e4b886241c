"""CPU test against ``tests/golden/mib_flow_reference.{json,npz}`` -- losses and final parameters the REFERENCE's own
``nnUNetTrainerMiB.run_training / run_iteration`` produced (oracle/make_goldens_mib.py: two tasks, with and without
``transfer_heads``).  Pins the oracle's MiB iteration, against which tests/test_cl_gpu.py checks the HIP trainer."""
import copy
import json

import numpy as np
import pytest
import torch

from oracle import losses as olosses, train as otrain
from oracle.unet import OracleGenericUNet
from lifelong_nnunet_amd.synthetic import make_patch_batch


@pytest.fixture(scope="module")
def ref(golden_dir):
    return (json.load(open(golden_dir + "/mib_flow_reference.json")), np.load(golden_dir + "/mib_flow_reference.npz"),
            np.load(golden_dir + "/trainer_reference.npz"))


def ref_batches(task_seed, n):
    out = []
    for i in range(n):
        data, tgts = make_patch_batch(2, (16, 16, 16), 2, seed=task_seed + i)
        out.append({"data": data, "target": tgts})
    return out


@pytest.mark.parametrize("key", ["init", "transfer"])
def test_oracle_mib_flow_equals_reference(ref, key):
    meta, arr, tarr = ref
    f = meta["mib_flow_" + key]
    init = {n[6:]: torch.from_numpy(tarr[n]) for n in tarr.files if n.startswith("init::")}
    net = OracleGenericUNet(1, 8, 3, 2)
    net.load_state_dict(init)
    opt = otrain.make_optimizer(net)
    w = olosses.ds_loss_weights(2)
    lA = [otrain.run_iteration(net, opt, b["data"], b["target"], w)[0] for b in ref_batches(f["seeds"]["taskA"], 4)[:2]]
    assert np.allclose(lA, f["lossesA"], rtol=1e-6), (lA, f["lossesA"])
    old = copy.deepcopy(net)                                # MiB.py:96 (stays in train mode; InstanceNorm has no running statistics)
    assert f["old_requires_grad"]                           # the reference neither freezes nor detaches it: only wasted work
    if not f["transfer_heads"]:                             # add_new_task(use_init=True): the new head starts from the initial head
        with torch.no_grad():
            for n, p in net.named_parameters():
                if n.startswith("seg_outputs."):
                    p.copy_(init[n])
    lB = []
    for b in ref_batches(f["seeds"]["taskB"], 4)[:3]:
        opt.zero_grad()
        out = net(b["data"])
        with torch.no_grad():
            out_o = old(b["data"])
        l = olosses.mib_loss(out, out_o, b["target"], w, f["alpha"], float(f["lkd"]))
        l.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 12)
        opt.step()
        lB.append(float(l.detach()))
    assert np.allclose(lB, f["lossesB"], rtol=1e-6), (lB, f["lossesB"])
    flat = torch.cat([dict(net.named_parameters())[n].detach().reshape(-1) for n in f["names"]]).numpy()
    exp = arr[f"mib_{key}::final_theta::sub"]
    assert np.linalg.norm(flat[::7] - exp) <= 1e-6 * np.linalg.norm(exp)
