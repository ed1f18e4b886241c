"""Generate tests/golden/plop_reference.{npz,json} by EXECUTING THE REFERENCE'S PLOP / POD CODE -- run in the build
container only (needs /root/reference):

    python -m oracle.make_goldens_plop

Same technique as oracle/make_goldens_trainers.py (reference classes imported through ``oracle.ref_shim``, instantiated
without their constructors, the multi-head base class' epoch loop replaced by a 3-line loop).  Executed verbatim:

  embeddings.local_POD / pod_embed                                   embeddings.py:3-41
  MultipleOutputLossPLOP.forward / _pseudo_label_loss                deep_supervision.py:248-334
  MultipleOutputLossPOD.forward                                      deep_supervision.py:362-381
  nnUNetTrainerPLOP.extract_max_entropy_and_thresholds               PLOP.py:114-172
  nnUNetTrainerPLOP.run_training / run_iteration / register_forward_hooks / reinitialize   PLOP.py:101-112,174-358
  nnUNetTrainerPOD.run_training / run_iteration / reinitialize       POD.py:58-96

The PLOP flows start from a CONFIDENT first head (the seg_outputs weights of the seeded initial state times 60): with the
0.001 threshold floor the reference ends up with (see oracle/plop.py) an unconfident toy network has no valid pseudo
label at all and the reference's loss is NaN from the first PLOP iteration on -- that case is recorded too
(``plop_flow_unconfident``).  Only DATA is written.  TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import copy
import json
import math
import os
import tempfile
import warnings
from collections import OrderedDict

import numpy as np
import torch

from . import losses, plop as oplop, ref_shim, train
from .make_goldens_trainers import OUT, PATCH, SUB, TOY_CTOR, batches, new_trainer, put
from .unet import OracleGenericUNet

BOOST = 60.0


class Counting:
    def __init__(self, items):
        self.items, self.n = items, 0

    def __iter__(self):
        return self

    def __next__(self):
        b = self.items[self.n % len(self.items)]
        self.n += 1
        return b


def main():
    os.makedirs(OUT, exist_ok=True)
    ref_shim.install()
    warnings.filterwarnings("ignore")
    _, RefMH = ref_shim.import_trainer("multihead", "nnUNetTrainerMultiHead")
    plop_mod, RefPLOP = ref_shim.import_trainer("plop", "nnUNetTrainerPLOP")
    pod_mod, RefPOD = ref_shim.import_trainer("pod", "nnUNetTrainerPOD")
    import nnunet_ext.training.loss_functions.deep_supervision as ds
    from nnunet_ext.training.loss_functions.embeddings import local_POD
    arrs, meta = {}, {}

    # ------------------------------------------------------------------ local_POD on random tensors
    g = torch.Generator().manual_seed(77)
    cases = []
    for i, (shape, scales) in enumerate([((2, 4, 6, 8, 8), 3), ((2, 3, 5, 12, 12), 3), ((2, 2, 3, 6, 6), 3), ((1, 8, 4, 16, 16), 4),
                                         ((2, 3, 2, 4, 4), 2), ((2, 16, 3, 8, 8), 3)]):
        a, b = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
        v = float(local_POD(a, b, scales))
        assert abs(v - oplop.local_pod(a, b, scales)) <= 1e-6 * abs(v), (shape, v, oplop.local_pod(a, b, scales))
        arrs[f"pod::a{i}"], arrs[f"pod::b{i}"] = a.numpy(), b.numpy()
        cases.append({"shape": list(shape), "scales": scales, "value": v})
    # the failure modes
    fails = {}
    try:
        local_POD(torch.zeros(2, 2, 2, 8, 6), torch.zeros(2, 2, 2, 8, 6), 3)
        fails["non_square"] = "ok"
    except Exception as e:       # noqa: BLE001
        fails["non_square"] = type(e).__name__
    try:
        local_POD(torch.zeros(2, 2, 2, 2, 2), torch.zeros(2, 2, 2, 2, 2), 3)
        fails["too_many_scales"] = "ok"
    except Exception as e:       # noqa: BLE001
        fails["too_many_scales"] = type(e).__name__
    meta["local_pod"] = {"cases": cases, "fails": fails}

    # ------------------------------------------------------------------ the PLOP loss on synthetic logits
    with ref_shim.cuda_as_cpu():
        g = torch.Generator().manual_seed(78)
        K, B = 3, 2
        dims = [(8, 8, 8), (4, 4, 4), (2, 2, 2)]
        x = [(torch.randn((B, K) + d, generator=g) * 2).requires_grad_(True) for d in dims]
        x_o = [torch.randn((B, K) + d, generator=g) * 6 for d in dims]
        y = [(torch.rand((B, 1) + d, generator=g) * 3).floor().clamp(max=2) * (torch.rand((B, 1) + d, generator=g) > 0.5) for d in dims]
        thr = {i: torch.tensor(t) for i, t in enumerate([[0.30, 0.20, 0.25], [0.35, 0.30, 0.20], [0.2, 0.2, 0.2]])}
        weights = np.array([4 / 7, 2 / 7, 0.0])
        interm = OrderedDict((f"l{i}", torch.randn(s, generator=g)) for i, s in enumerate([(2, 4, 8, 8, 8), (2, 8, 4, 4, 4), (2, 3, 8, 8, 8)]))
        old = OrderedDict((k, v + 0.3 * torch.randn(v.shape, generator=g)) for k, v in interm.items())
        per_level = []
        orig_pl = ds.MultipleOutputLossPLOP._pseudo_label_loss

        def rec_pl(self, x_, xo_, y_, idx):
            v = orig_pl(self, x_, xo_, y_, idx)
            per_level.append([idx, float(v)])
            return v
        ds.MultipleOutputLossPLOP._pseudo_label_loss = rec_pl
        try:
            L = ds.MultipleOutputLossPLOP(K - 1, 0.01, 3, weights)
            L.update_plop_params(old, interm, thr, torch.log(torch.tensor(float(K))))
            val = L(x, x_o, y)
            val.backward()
        finally:
            ds.MultipleOutputLossPLOP._pseudo_label_loss = orig_pl
        o_val = oplop.plop_loss([t.detach() for t in x], x_o, y, weights, interm, old, thr, math.log(K), 0.01, 3)
        assert abs(float(val) - float(o_val)) <= 1e-6 * abs(float(val)), (float(val), float(o_val))
        for i in range(3):
            arrs[f"loss::x{i}"], arrs[f"loss::xo{i}"], arrs[f"loss::y{i}"] = x[i].detach().numpy(), x_o[i].numpy(), y[i].numpy()
        for i in range(2):
            arrs[f"loss::dx{i}"] = x[i].grad.numpy()
        assert x[2].grad is None
        for k_ in interm:
            arrs[f"loss::h_{k_}"], arrs[f"loss::ho_{k_}"] = interm[k_].numpy(), old[k_].numpy()
        meta["plop_loss"] = {"value": float(val), "per_level": per_level, "thresholds": [thr[i].tolist() for i in range(3)],
                             "weights": weights.tolist(), "pod_lambda": 0.01, "scales": 3, "layers": list(interm.keys()),
                             "dist": float(val) - sum(weights[i] * v for i, v in per_level)}
        # POD loss = base + dist
        base = plop_mod.DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False}, {}) if hasattr(plop_mod, "DC_and_CE_loss") \
            else pod_mod.DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False}, {})
        LP = ds.MultipleOutputLossPOD(base, weights, 0.01, 3)
        LP.update_plop_params(old, interm)
        xs = [t.detach() for t in x]
        pv = float(LP(xs, y))
        bv = float(ref_shim.MultipleOutputLoss2(base, weights)(xs, y))
        assert abs(pv - float(oplop.pod_loss(bv, interm, old, 0.01, 3))) <= 1e-6 * abs(pv)
        meta["pod_loss"] = {"value": pv, "base": bv}

    # ------------------------------------------------------------------ trainer flows
    def light_run_training(self, task, output_folder=None, build_folder=True):
        """Stand-in for nnUNetTrainerMultiHead.run_training (MH.py:520-596: epoch loop + files): N training iterations."""
        if self.task != task:             # MH.py:541-545
            self.reinitialize(task)
            self.task = task
        if task not in self.mh_network.heads:       # MH.py:551-552
            self.mh_network.add_new_task(task, use_init=not self.transfer_heads)
        self.network = self.mh_network.assemble_model(task)        # MH.py:566
        self.network.train()
        out = [float(self.run_iteration(self.tr_gen, True)) for _ in range(self.num_batches_per_epoch)]
        self.loop_losses = out
        return out

    def light_reinitialize(self, task, print_loss_info=True):
        """Stand-in for nnUNetTrainerMultiHead.reinitialize (MH.py:458-518: new data loaders from files)."""
        self.tr_gen = self.gens[task]

    def fresh_mh():
        import importlib
        import nnunet_ext.network_architecture.MultiHead_Module as mhm
        importlib.reload(mhm)
        return mhm.MultiHead_Module(OracleGenericUNet, "seg_outputs", "taskA", None, *TOY_CTOR)

    pods = []
    orig_local_pod = ds.local_POD

    def rec_local_pod(h, h_old, scales):
        v = orig_local_pod(h, h_old, scales)
        pods.append(float(v))
        return v

    def flow(RefCls, mod, boost, tasks, pod_trainer):
        with tempfile.TemporaryDirectory() as td, ref_shim.cuda_as_cpu():
            torch.manual_seed(12345)
            mh = fresh_mh()
            net = mh.model
            if boost != 1.0:
                with torch.no_grad():
                    for n_, p_ in net.named_parameters():
                        if n_.startswith("seg_outputs."):
                            p_.mul_(boost)
                mh.update_after_iteration()
            names = [n for n, _ in net.named_parameters()]
            tr = new_trainer(RefCls, net, "taskA", num_batches_per_epoch=2, mh_network=mh, transfer_heads=False, split_gpu=False,
                             use_vit=False, num_classes=3, pod_lambda=0.01, scales=3, old_interm_results=dict(), interm_results=dict(),
                             switched=False, network_name="3d_fullres",
                             already_trained_on={"0": {"finished_training_on": []}})
            if not pod_trainer:
                tr.thresholds, tr.max_entropy = None, dict()          # PLOP.py:74 (the order in the constructor)
            tr.initialize_optimizer_and_scheduler()
            tr._update_loss_after_plans_change([[2, 2, 2]] * 2, PATCH)
            tr.loss_orig = copy.deepcopy(tr.loss)                      # PLOP.py:89
            if pod_trainer:
                base = pod_mod.DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False}, {})
                tr.loss_plop = ds.MultipleOutputLossPOD(base, tr.ds_loss_weights, tr.pod_lambda, tr.scales)     # POD.py:48-51
            else:
                tr.loss_plop = ds.MultipleOutputLossPLOP(tr.num_classes - 1, tr.pod_lambda, tr.scales, tr.ds_loss_weights)   # PLOP.py:94-97
            gens = {t: Counting(batches(seed, 8)) for t, seed in tasks}
            tr.gens = gens
            tr.tr_gen = gens["taskA"]
            out = {"tasks": [t for t, _ in tasks], "seeds": {t: s for t, s in tasks}, "boost": boost, "names": names}
            for t, _ in tasks:
                del pods[:], order_seen[:], level_seen[:]
                tr.run_training(t, td)
                out["order_" + t] = [list(o) for o in order_seen]
                out["pseudo_levels_" + t] = [list(v) for v in level_seen]
                out["losses_" + t] = tr.loop_losses
                out["consumed_after_" + t] = {k: g_.n for k, g_ in gens.items()}
                out["pods_" + t] = list(pods)
                if t != "taskA":
                    out["layers"] = [n for n, m in tr.network_old.named_modules() if 'conv.Conv' in str(type(m))]
                if not pod_trainer and t == "taskB":
                    out["thresholds_reset"] = [tr.thresholds == dict(), tr.max_entropy is None]
            out["hook_counts"] = {"network": len(net.seg_outputs[0]._forward_hooks), "network_old": len(tr.network_old.seg_outputs[0]._forward_hooks)}
            return out, dict(net.named_parameters()), names, {k: v.clone() for k, v in mh.state_init.items()} if hasattr(mh, "state_init") else None

    orig_run_training, orig_reinitialize = RefMH.run_training, RefMH.reinitialize
    RefMH.run_training, RefMH.reinitialize = light_run_training, light_reinitialize
    ds.local_POD = rec_local_pod
    order_seen, level_seen = [], []
    orig_upd_plop, orig_upd_pod = ds.MultipleOutputLossPLOP.update_plop_params, ds.MultipleOutputLossPOD.update_plop_params

    def rec_upd_plop(self, old_interm_results, interm_results, thresholds, max_entropy):
        order_seen.append(list(old_interm_results.keys()))
        return orig_upd_plop(self, old_interm_results, interm_results, thresholds, max_entropy)

    def rec_upd_pod(self, old_interm_results, interm_results):
        order_seen.append(list(old_interm_results.keys()))
        return orig_upd_pod(self, old_interm_results, interm_results)
    ds.MultipleOutputLossPLOP.update_plop_params, ds.MultipleOutputLossPOD.update_plop_params = rec_upd_plop, rec_upd_pod
    orig_pl2 = ds.MultipleOutputLossPLOP._pseudo_label_loss

    def rec_pl2(self, x_, xo_, y_, idx):
        v = orig_pl2(self, x_, xo_, y_, idx)
        level_seen.append([idx, float(v)])
        return v
    ds.MultipleOutputLossPLOP._pseudo_label_loss = rec_pl2
    # thresholds: recorded from inside the reference's extraction
    thr_seen = []
    orig_extract = RefPLOP.extract_max_entropy_and_thresholds

    def rec_extract(self):
        orig_extract(self)
        thr_seen.append({"max_entropy": float(self.max_entropy), "thresholds": {str(k): v.tolist() for k, v in self.thresholds.items()}})
    RefPLOP.extract_max_entropy_and_thresholds = rec_extract
    try:
        out, theta, names, _ = flow(RefPLOP, plop_mod, BOOST, [("taskA", 8000), ("taskB", 8100), ("taskC", 8200)], False)
        out["extracted"] = list(thr_seen)
        put(arrs, "plop::final_theta", theta, names)
        meta["plop_flow"] = out
        del thr_seen[:]
        out, theta, names, _ = flow(RefPLOP, plop_mod, 1.0, [("taskA", 8000), ("taskB", 8100)], False)
        meta["plop_flow_unconfident"] = {k: out[k] for k in ("losses_taskA", "losses_taskB", "pods_taskB", "boost")}
        out, theta, names, _ = flow(RefPOD, pod_mod, 1.0, [("taskA", 8000), ("taskB", 8100), ("taskC", 8200)], True)
        put(arrs, "pod::final_theta", theta, names)
        meta["pod_flow"] = out
    finally:
        RefMH.run_training, RefMH.reinitialize = orig_run_training, orig_reinitialize
        ds.local_POD = orig_local_pod
        ds.MultipleOutputLossPLOP.update_plop_params, ds.MultipleOutputLossPOD.update_plop_params = orig_upd_plop, orig_upd_pod
        ds.MultipleOutputLossPLOP._pseudo_label_loss = orig_pl2
        RefPLOP.extract_max_entropy_and_thresholds = orig_extract
    meta["sub"] = SUB

    np.savez_compressed(os.path.join(OUT, "plop_reference.npz"), **arrs)
    with open(os.path.join(OUT, "plop_reference.json"), "w") as f:
        json.dump(meta, f, indent=1)
    for fn in ("plop_reference.npz", "plop_reference.json"):
        print("wrote", os.path.join(OUT, fn), os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
