"""Generate tests/golden/mib_flow_reference.{npz,json} by EXECUTING THE REFERENCE'S MiB TRAINER -- run in the build container
only (needs /root/reference):

    python -m oracle.make_goldens_mib

Same technique as oracle/make_goldens_trainers.py.  Executed verbatim: nnUNetTrainerMiB.run_training / reinitialize /
run_iteration (MiB.py:71-182) around the reference's MultiHead_Module and MultipleOutputLossMiB; task A (2 iterations, the
plain loss), task B (3 iterations: forward of the current and of the deep-copied old model on the same batch, CE + unbiased
knowledge distillation, clip 12, SGD).  Only DATA is written.  TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import copy
import json
import os
import tempfile
import warnings

import numpy as np
import torch

from . import ref_shim
from .make_goldens_trainers import OUT, PATCH, SUB, TOY_CTOR, batches, new_trainer, put
from .unet import OracleGenericUNet


def main():
    ref_shim.install()
    warnings.filterwarnings("ignore")
    _, RefMH = ref_shim.import_trainer("multihead", "nnUNetTrainerMultiHead")
    mib_mod, RefMiB = ref_shim.import_trainer("mib", "nnUNetTrainerMiB")
    import nnunet_ext.training.loss_functions.deep_supervision as ds
    arrs, meta = {}, {}

    def light_run_training(self, task, output_folder=None, build_folder=True):
        """Stand-in for nnUNetTrainerMultiHead.run_training (MH.py:520-596: epoch loop + files): N training iterations."""
        if self.task != task:
            self.reinitialize(task)
            self.task = task
        if task not in self.mh_network.heads:
            self.mh_network.add_new_task(task, use_init=not self.transfer_heads)
        self.network = self.mh_network.assemble_model(task)
        self.network.train()
        self.loop_losses = [float(self.run_iteration(self.tr_gen, True)) for _ in range(self.num_batches_per_epoch)]
        return self.loop_losses

    def light_reinitialize(self, task, print_loss_info=True):
        self.tr_gen = self.gens[task]

    def fresh_mh():
        import importlib
        import nnunet_ext.network_architecture.MultiHead_Module as mhm
        importlib.reload(mhm)
        return mhm.MultiHead_Module(OracleGenericUNet, "seg_outputs", "taskA", None, *TOY_CTOR)

    orig_run_training, orig_reinitialize = RefMH.run_training, RefMH.reinitialize
    RefMH.run_training, RefMH.reinitialize = light_run_training, light_reinitialize
    try:
        for transfer in (False, True):
            with tempfile.TemporaryDirectory() as td, ref_shim.cuda_as_cpu():
                torch.manual_seed(12345)
                mh = fresh_mh()
                net = mh.model
                names = [n for n, _ in net.named_parameters()]
                tr = new_trainer(RefMiB, net, "taskA", num_batches_per_epoch=2, mh_network=mh, transfer_heads=transfer, split_gpu=False,
                                 use_vit=False, alpha=1.0, lkd=10, already_trained_on={"0": {"finished_training_on": []}})
                tr.initialize_optimizer_and_scheduler()
                tr._update_loss_after_plans_change([[2, 2, 2]] * 2, PATCH)
                tr.loss_orig = copy.deepcopy(tr.loss)                                    # MiB.py:65
                tr.loss_mib = ds.MultipleOutputLossMiB(tr.alpha, tr.lkd, tr.ds_loss_weights)   # MiB.py:70-73
                tr.gens = {"taskA": iter(batches(9000, 4)), "taskB": iter(batches(9100, 4))}
                tr.tr_gen = tr.gens["taskA"]
                tr.run_training("taskA", td)
                lA = tr.loop_losses
                tr.num_batches_per_epoch = 3
                tr.run_training("taskB", td)
                key = "transfer" if transfer else "init"
                meta["mib_flow_" + key] = {"lossesA": lA, "lossesB": tr.loop_losses, "names": names, "alpha": 1.0, "lkd": 10,
                                           "seeds": {"taskA": 9000, "taskB": 9100}, "transfer_heads": transfer,
                                           "old_requires_grad": any(p.requires_grad for p in tr.network_old.parameters())}
                put(arrs, f"mib_{key}::final_theta", dict(net.named_parameters()), names)
    finally:
        RefMH.run_training, RefMH.reinitialize = orig_run_training, orig_reinitialize
    meta["sub"] = SUB
    np.savez_compressed(os.path.join(OUT, "mib_flow_reference.npz"), **arrs)
    with open(os.path.join(OUT, "mib_flow_reference.json"), "w") as f:
        json.dump(meta, f, indent=1)
    for fn in ("mib_flow_reference.npz", "mib_flow_reference.json"):
        print("wrote", os.path.join(OUT, fn), os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
