"""Generates tests/golden/checkpoint_reference.{npz,json}: a checkpoint as the REFERENCE produces it.

``nnUNetTrainerMultiHead.save_checkpoint`` (MH.py:1164-1197) stores ``self.mh_network.state_dict()`` -- the state of the
reference's ``MultiHead_Module`` (``model.* / body.* / heads.<task>.*``) -- next to ``optimizer.state_dict()`` of a
``torch.optim.SGD`` (upstream ``NetworkTrainer.save_checkpoint``).  Here the REFERENCE class
(``/root/reference/nnunet_ext/network_architecture/MultiHead_Module.py``, imported, executed) wraps the oracle network:
two heads with different weights, three real optimisation steps on head B (so the momentum buffers are not zero), then
``state_dict()`` of the module and of the optimiser are written as arrays + key lists.
tests/test_host_logic.py::test_load_checkpoint_written_by_the_reference_classes loads it with ``load_checkpoint_ram``.

    python -m oracle.make_goldens_checkpoint        (in the build container; /root/reference is not on the GPU box)

Only DATA is written (npz / json): no reference source or bytecode is copied."""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from . import losses, train
from .make_goldens import OUT, install_shim
from .unet import OracleGenericUNet


def main():
    install_shim()
    from nnunet_ext.network_architecture.MultiHead_Module import MultiHead_Module
    from lifelong_nnunet_amd.synthetic import make_patch_batch
    torch.manual_seed(21)
    mh = MultiHead_Module(OracleGenericUNet, "seg_outputs", "taskA", None, 1, 8, 3, 2)
    mh.add_new_task("taskB", use_init=True)                                  # MHM.py:435-458
    net = mh.assemble_model("taskB")                                         # MHM.py:326-377: head B active
    opt = train.make_optimizer(net)
    w = losses.ds_loss_weights(2)
    for i in range(3):
        data, tgts = make_patch_batch(2, (16, 16, 16), 2, seed=900 + i)
        train.run_iteration(net, opt, data, tgts, w)
        mh.update_after_iteration()                                          # MHM.py:139-157
    sd = mh.state_dict()
    arrays = {"sd::" + k: v.detach().cpu().numpy() for k, v in sd.items()}
    osd = opt.state_dict()
    for idx, st in osd["state"].items():
        arrays[f"opt::{idx}"] = st["momentum_buffer"].detach().cpu().numpy()
    trainable = [n for n, p in net.named_parameters() if p.requires_grad]
    x, _ = make_patch_batch(2, (16, 16, 16), 2, seed=950)
    logits = {}
    for t in ("taskA", "taskB"):
        m = mh.assemble_model(t)
        m.eval()
        with torch.no_grad():
            logits[t] = m(x)[0].numpy()
        arrays["logits::" + t] = logits[t]
    mh.assemble_model("taskB")
    group = dict(osd["param_groups"][0])
    meta = {"state_dict_keys": list(sd.keys()), "heads": list(mh.heads.keys()), "active_task": str(mh.active_task),
            "optimizer_param_names": trainable, "optimizer_state_indices": sorted(int(i) for i in osd["state"].keys()),
            "optimizer_group": {k: group[k] for k in ("lr", "momentum", "weight_decay", "nesterov", "dampening")},
            "optimizer_params": list(group["params"]), "epoch": 3, "probe_seed": 950, "ctor": [1, 8, 3, 2],
            "heads_differ": float(np.abs(logits["taskA"] - logits["taskB"]).max())}
    np.savez_compressed(os.path.join(OUT, "checkpoint_reference.npz"), **arrays)
    with open(os.path.join(OUT, "checkpoint_reference.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote checkpoint_reference.*:", len(sd), "state-dict tensors,", len(osd["state"]), "momentum buffers; heads differ by",
          meta["heads_differ"])


if __name__ == "__main__":
    main()
