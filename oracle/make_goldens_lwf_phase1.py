"""Generates tests/golden/lwf_phase1_reference.{npz,json}: LwF PHASE 1 (the head-only warm-up with a frozen body,
lwf/nnUNetTrainerLWF.py:189-201) executed by the REFERENCE's own classes.

Phase 1 is ``self.network = self.mh_network.assemble_model(task, freeze_body=True)`` (MHM.py:326-395: the body's
``requires_grad`` goes False), ``self.loss = self.loss_orig`` and upstream's epoch loop, whose iterations are
``nnUNetTrainerLWF.run_iteration`` (LWF.py:298-370) with ``freeze_run == True`` -> the plain branch (:303-307) ->
``nnUNetTrainerMultiHead.run_iteration`` (MH.py:598-656).  Everything named is in the reference tree and runs verbatim
through oracle/ref_shim (as in make_goldens_trainers.py); only upstream's epoch loop is a 3-iteration ``for``.
Recorded: the state phase 1 starts from (model + both heads; the last decoder block's InstanceNorm weight is scaled so that
the head's gradient norm exceeds 12 -- upstream's plain iteration does not clip), which parameters are frozen, the three
phase-1 losses and gradient norms, the final parameters (body unchanged bit for bit, head B moved), batches consumed.

    python -m oracle.make_goldens_lwf_phase1        (build container only)

Only DATA is written.  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import json
import os
import tempfile
import warnings

import numpy as np
import torch

from . import ref_shim
from .make_goldens_trainers import OUT, PATCH, TOY_CTOR, batches, new_trainer, put
from .unet import OracleGenericUNet


def main():
    ref_shim.install()
    warnings.filterwarnings("ignore")
    lwf_mod, RefLWF = ref_shim.import_trainer("lwf", "nnUNetTrainerLWF")
    from nnunet_ext.training.loss_functions.deep_supervision import MultipleOutputLossLWF
    import importlib
    import nnunet_ext.network_architecture.MultiHead_Module as mhm
    importlib.reload(mhm)                                  # mutable default arguments of the splitting helper (MHM.py:159-160)

    class Counting:
        def __init__(self, items):
            self.items, self.n = items, 0

        def __iter__(self):
            return self

        def __next__(self):
            b = self.items[self.n % len(self.items)]
            self.n += 1
            return b

    arrs, meta = {}, {}
    with tempfile.TemporaryDirectory(), ref_shim.cuda_as_cpu():
        torch.manual_seed(4242)
        mh = mhm.MultiHead_Module(OracleGenericUNet, "seg_outputs", "taskA", None, *TOY_CTOR)
        net = mh.model
        names = [n for n, _ in net.named_parameters()]
        tr = new_trainer(RefLWF, net, "taskA", num_batches_per_epoch=3, mh_network=mh, transfer_heads=False, freeze_run=False,
                         do_val=False, use_vit=False, ViT_task_specific_ln=False, batch_idx=0, lwf_temperature=2.0,
                         already_trained_on={"0": {"finished_training_on": []}})
        tr.initialize_optimizer_and_scheduler()
        tr._update_loss_after_plans_change([[2, 2, 2]] * 2, PATCH)
        tr.loss_orig = tr.loss
        base = lwf_mod.DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False}, {})
        tr.LwFloss = MultipleOutputLossLWF(base, tr.ds_loss_weights, list(), list(), tr.lwf_temperature)
        gA = Counting(batches(8000, 2))
        tr.network.train()
        meta["lossesA"] = [float(tr.run_iteration(gA, True)) for _ in range(2)]
        mh.add_new_task("taskB", use_init=True)                                          # MH.py:551-552
        # ---- phase 1 (LWF.py:189-201)
        tr.freeze_run = True
        tr.network = mh.assemble_model("taskB", freeze_body=True)
        # make the (frozen) last decoder block's output large so that the head's gradient norm exceeds 12: the upstream
        # iteration phase 1 runs does NOT clip -- a trainer that clips at 12 there ends on different weights
        with torch.no_grad():
            dict(tr.network.named_parameters())["conv_blocks_localization.1.1.blocks.0.instnorm.weight"].mul_(150.0)
        before = {n: p.detach().clone() for n, p in net.named_parameters()}
        for n, p in net.named_parameters():
            arrs["pre::" + n] = p.detach().numpy().copy()
        for t in ("taskA", "taskB"):
            for n, p in mh.heads[t].named_parameters():
                arrs[f"prehead::{t}::{n}"] = p.detach().numpy().copy()
        tr.initialize_optimizer_and_scheduler()                                           # fresh momentum: the flow starts here
        tr.loss = tr.loss_orig
        tr.task = "taskB"
        meta["frozen"] = [n for n, p in tr.network.named_parameters() if not p.requires_grad]
        gB = Counting(batches(9000, 3))
        tr.network.train()
        meta["losses_phase1"], meta["grad_norms_phase1"] = [], []
        for _ in range(3):
            meta["losses_phase1"].append(float(tr.run_iteration(gB, True)))
            gs = [p.grad for p in tr.network.parameters() if p.grad is not None]
            meta["grad_norms_phase1"].append(float(torch.sqrt(sum((g.double() ** 2).sum() for g in gs))))
        assert min(meta["grad_norms_phase1"]) > 24.0, meta["grad_norms_phase1"]
        meta["batches_consumed_phase1"] = gB.n
        meta["batch_idx"] = tr.batch_idx
        after = dict(tr.network.named_parameters())
        meta["body_unchanged"] = bool(all(torch.equal(before[n], after[n].detach()) for n in meta["frozen"]))
        meta["head_moved"] = float(max((before[n] - after[n].detach()).abs().max() for n in names if n not in meta["frozen"]))
        put(arrs, "phase1::final_theta", after, names)
        put(arrs, "phase1::headB", dict(mh.heads["taskB"].named_parameters()), [n for n, _ in mh.heads["taskB"].named_parameters()])
    meta["names"] = names
    meta["seeds"] = {"taskA": 8000, "taskB": 9000}
    np.savez_compressed(os.path.join(OUT, "lwf_phase1_reference.npz"), **arrs)
    with open(os.path.join(OUT, "lwf_phase1_reference.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote lwf_phase1_reference.*", {k: meta[k] for k in ("lossesA", "losses_phase1", "batches_consumed_phase1", "batch_idx",
                                                                 "body_unchanged", "head_moved")}, "frozen:", len(meta["frozen"]), "of", len(names))


if __name__ == "__main__":
    main()
