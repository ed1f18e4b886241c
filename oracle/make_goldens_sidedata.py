"""Generate tests/golden/sidedata_reference.{npz,json} by EXECUTING THE REFERENCE'S ``run_training`` TAILS that persist a finished task's
side data -- run in the build container only (needs /root/reference):

    python -m oracle.make_goldens_sidedata

What is executed verbatim (through ``oracle.ref_shim``, trainers created with ``object.__new__`` exactly as in
``oracle.make_goldens_trainers``; same seeds, same toy network, same synthetic batches -- the in-memory Fisher / theta* of this run
ARE the ``ewc::fisherA`` / ``ewc::paramsA`` of tests/golden/trainer_reference.npz, which the tests cross-check):

  nnUNetTrainerEWC.run_training  EWC.py:179-230 -- in particular :205-228: ``write_pickle(self.fisher, ewc_data/fisher_values.pkl)``,
      ``write_pickle(self.params, ewc_data/param_values.pkl)``, ``fisher_at`` / ``params_at`` entered into ``already_trained_on``,
      ``save_json(already_trained_on, <ext>_trained_on.pkl)``
  nnUNetTrainerRW.run_training   RW.py:150-208 + ``save_f_p_s_values`` :267-300 (``rw_data/{fisher,param,score}_values.pkl``)

The CONTENT of the files the reference wrote is the fixture: every pickle is read back with plain ``pickle.load`` (it is a dictionary
task -> parameter name -> CPU tensor) and stored as arrays ``<method>::<file>::<task>::<name>`` in the .npz; the JSON the reference
writes under a ``.pkl`` name is stored as parsed objects in the .json together with the layout (which file sits where).  The raw
pickles themselves are not committed because pickling a torch tensor embeds the storage's memory address as a key -- two runs of this
script give files that differ in those bytes; tests/test_host_logic.py re-creates the files (``pickle.dump`` of the same dictionaries =
what ``write_pickle`` does) and lets the product's trainers load them.  TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import contextlib
import json
import os
import pickle
import shutil
import warnings

import numpy as np
import torch

from . import ref_shim
from .make_goldens_trainers import PATCH, TOY_CTOR, batches, new_trainer
from .unet import OracleGenericUNet

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
WORK = "/tmp/lnn_sidedata_reference_run"      # a FIXED scratch path: it ends up inside the JSON the reference writes


@contextlib.contextmanager
def workdir():
    shutil.rmtree(WORK, ignore_errors=True)
    os.makedirs(WORK)
    try:
        yield WORK
    finally:
        shutil.rmtree(WORK, ignore_errors=True)


def main():
    os.makedirs(OUT, exist_ok=True)
    arrs = {}
    ref_shim.install()
    warnings.filterwarnings("ignore")
    _, RefMH = ref_shim.import_trainer("multihead", "nnUNetTrainerMultiHead")
    ewc_mod, RefEWC = ref_shim.import_trainer("ewc", "nnUNetTrainerEWC")
    rw_mod, RefRW = ref_shim.import_trainer("rw", "nnUNetTrainerRW")
    from nnunet_ext.training.loss_functions.deep_supervision import MultipleOutputLossEWC, MultipleOutputLossRW

    def light_run_training(self, task, output_folder=None, build_folder=True):
        """Stand-in for nnUNetTrainerMultiHead.run_training (MH.py:520-596: epoch loop + files): N training iterations."""
        if self.task != task:
            self.reinitialize(task)
            self.task = task
        if task not in self.mh_network.heads:
            self.mh_network.add_new_task(task, use_init=not self.transfer_heads)
        self.network = self.mh_network.assemble_model(task)
        self.network.train()
        out = [float(self.run_iteration(self.tr_gen, True)) for _ in range(self.num_batches_per_epoch)]
        self.already_trained_on[str(self.fold)]['finished_training_on'].append(task)
        return out

    def light_reinitialize(self, task, print_loss_info=True):
        self.tr_gen = self.gens[task]

    def fresh_mh():
        import importlib
        import nnunet_ext.network_architecture.MultiHead_Module as mhm
        importlib.reload(mhm)
        return mhm.MultiHead_Module(OracleGenericUNet, "seg_outputs", "taskA", None, *TOY_CTOR)

    def keep(td, sub, files, trained_on_name, meta_key, meta, already):
        """Read back what the reference wrote under ``td``: pickles -> arrays, the trained-on JSON -> object (paths relative)."""
        for f in files:
            with open(os.path.join(td, sub, f), "rb") as fh:
                d = pickle.load(fh)                       # plain pickle of {task: {name: tensor}}
            for task, per in d.items():
                for name, t in per.items():
                    assert isinstance(t, torch.Tensor) and t.device.type == "cpu"
                    arrs[f"{meta_key}::{f}::{task}::{name}"] = t.detach().numpy().copy()
        with open(os.path.join(td, trained_on_name)) as fh:
            on_disk = json.load(fh)                       # save_json under a .pkl name (EWC.py:224)
        rel = lambda o: json.loads(json.dumps(o).replace(td + os.sep, ""))
        assert rel(on_disk) == rel(already)
        meta[meta_key] = {"already_trained_on": rel(already), "files": [os.path.join(sub, f) for f in files],
                          "trained_on_file": trained_on_name, "trained_on_file_is_json": True}

    meta = {}
    orig = RefMH.run_training, RefMH.reinitialize
    RefMH.run_training, RefMH.reinitialize = light_run_training, light_reinitialize
    try:
        # ------------------------------------------------------------------ EWC: task A, then the tail of run_training writes the files
        with workdir() as td, ref_shim.cuda_as_cpu():
            torch.manual_seed(12345)
            mh = fresh_mh()
            net = mh.model
            names = [n for n, _ in net.named_parameters()]
            tr = new_trainer(RefEWC, net, "taskA", ewc_lambda=0.4, fisher=dict(), params=dict(), num_batches_per_epoch=3,
                             mh_network=mh, transfer_heads=False,
                             already_trained_on={"0": {"finished_training_on": [], "fisher_at": None, "params_at": None}},
                             ewc_data_path=os.path.join(td, "ewc_data"), trained_on_path=td, extension="ewc", output_folder=td)
            tr.update_init_args = lambda: None
            tr.save_init_args = lambda *a, **k: None
            tr.initialize_optimizer_and_scheduler()
            tr._update_loss_after_plans_change([[2, 2, 2]] * 2, PATCH)
            base = ewc_mod.DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False}, {})
            tr.loss = MultipleOutputLossEWC(base, tr.ds_loss_weights, tr.ewc_lambda, tr.fisher, tr.params, tr.network.named_parameters())
            tr.gens = {"taskA": iter(batches(1000, 6))}
            tr.tr_gen = tr.gens["taskA"]
            tr.run_training("taskA", td)
            assert tr.already_trained_on["0"]["fisher_at"] == os.path.join(td, "ewc_data", "fisher_values.pkl")
            keep(td, "ewc_data", ["fisher_values.pkl", "param_values.pkl"], "ewc_trained_on.pkl", "ewc", meta, tr.already_trained_on)
            meta["ewc"]["names"] = names
            meta["ewc"]["fisher_l2"] = {n: float(tr.fisher["taskA"][n].double().norm()) for n in names}
            meta["ewc"]["params_l2"] = {n: float(tr.params["taskA"][n].double().norm()) for n in names}

        # ------------------------------------------------------------------ RW: task A
        with workdir() as td, ref_shim.cuda_as_cpu():
            torch.manual_seed(12345)
            mh = fresh_mh()
            net = mh.model
            tr = new_trainer(RefRW, net, "taskA", rw_lambda=0.4, alpha=0.9, fisher_update_after=2, fisher=dict(), params=dict(),
                             mh_network=mh, transfer_heads=False, scores=dict(), num_batches_per_epoch=5, prev_param=None,
                             prev_fisher=None, count=0,
                             already_trained_on={"0": {"finished_training_on": [], "fisher_at": None, "params_at": None, "scores_at": None}},
                             rw_data_path=os.path.join(td, "rw_data"), trained_on_path=td, extension="rw", output_folder=td)
            tr.update_init_args = lambda: None
            tr.save_init_args = lambda *a, **k: None
            tr.initialize_optimizer_and_scheduler()
            tr._update_loss_after_plans_change([[2, 2, 2]] * 2, PATCH)
            base = rw_mod.DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False}, {})
            tr.loss = MultipleOutputLossRW(base, tr.ds_loss_weights, tr.rw_lambda, tr.fisher, tr.params, tr.scores, tr.network.named_parameters())
            tr.gens = {"taskA": iter(batches(3000, 5))}
            tr.tr_gen = tr.gens["taskA"]
            tr.run_training("taskA", td)
            keep(td, "rw_data", ["fisher_values.pkl", "param_values.pkl", "score_values.pkl"], "rw_trained_on.pkl", "rw", meta, tr.already_trained_on)
            gn = list(tr.fisher["taskA"].keys())
            meta["rw"]["stat_names"] = gn
            meta["rw"]["fisher_l2"] = {n: float(tr.fisher["taskA"][n].double().norm()) for n in gn}
            meta["rw"]["scores_l2"] = {n: float(tr.scores["taskA"][n].double().norm()) for n in gn}
    finally:
        RefMH.run_training, RefMH.reinitialize = orig
    np.savez_compressed(os.path.join(OUT, "sidedata_reference.npz"), **arrs)
    with open(os.path.join(OUT, "sidedata_reference.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote sidedata_reference.{npz,json}:", len(arrs), "arrays")


if __name__ == "__main__":
    main()
