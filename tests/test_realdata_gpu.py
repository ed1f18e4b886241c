"""SURVEY.md 8(f) rank 4 + rank 3 tail on the HIP path (-m gpu): trainers fed from nnU-Net-PREPROCESSED FOLDERS
(``dataloading.PreprocessedDataProvider``: load_dataset / do_split / DataLoader3D, REH.py:105-164), then
``nnUNetTrainerMultiHead.validate`` (MH.py:1052-1135: every head, whole validation cases by tiled inference) and the
evaluator's per-subject dictionary (evaluator2.py:60-109).  The folders are written by the test in the upstream layout
(``<case>.npz`` with key ``data`` = image channels + segmentation, ``<case>.pkl`` with ``class_locations``)."""
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import evaluation as oev, inference as oinf, losses as olosses, train as otrain  # noqa: E402
from oracle.unet import OracleGenericUNet  # noqa: E402
from lifelong_nnunet_amd import get_trainer_class  # noqa: E402
from lifelong_nnunet_amd.dataloading import PreprocessedDataProvider  # noqa: E402

DEV = "cuda:0"
PLANS = {"patch_size": (16, 32, 16), "batch_size": 2, "num_pool": 2, "base_num_features": 8, "num_classes": 3,
         "num_input_channels": 1}


def _write_task(root, task, n, seed):
    folder = os.path.join(root, task, "nnUNetData_plans_v2.1_stage0")
    os.makedirs(folder)
    rng = np.random.RandomState(seed)
    for i in range(n):
        shape = (20 + 2 * (i % 3), 36 + (i % 2) * 4, 18 + (i % 4))
        seg = np.zeros(shape, dtype=np.float32)
        c = [int(s * (0.35 + 0.3 * rng.rand())) for s in shape]
        seg[c[0] - 3:c[0] + 3, c[1] - 5:c[1] + 5, c[2] - 3:c[2] + 3] = 1
        seg[c[0] - 1:c[0] + 2, c[1] - 2:c[1] + 2, c[2] - 1:c[2] + 2] = 2
        img = (rng.randn(1, *shape) * 0.5 + seg[None] * 1.5).astype(np.float32)      # intensities carry the labels
        np.savez(os.path.join(folder, f"{task}_{i:03d}.npz"), data=np.concatenate([img, seg[None]], 0))
        pickle.dump({"class_locations": {k: np.argwhere(seg == k) for k in (1, 2)}, "size_after_resampling": shape},
                    open(os.path.join(folder, f"{task}_{i:03d}.pkl"), "wb"))
    return folder


class _Recording:
    """Wraps the provider's generators so that the test sees the exact batches the trainer consumed."""

    def __init__(self, prov):
        self.prov, self.seen = prov, []

    def __getattr__(self, name):
        return getattr(self.prov, name)

    def _wrap(self, gen):
        outer = self

        class G:
            def __iter__(self_):
                return self_

            def __next__(self_):
                b = next(gen)
                outer.seen.append(b)
                return b
        return G()

    def generator_for(self, dataset, plans, split="train"):
        return self._wrap(self.prov.generator_for(dataset, plans, split))

    def __call__(self, task, split, plans):
        return self._wrap(self.prov(task, split, plans))


def test_rehearsal_from_preprocessed_folders_trains_and_validates(tmp_path):
    root = str(tmp_path)
    fa, fb = _write_task(root, "Task900_ToyA", 10, 1), _write_task(root, "Task901_ToyB", 10, 2)
    prov = _Recording(PreprocessedDataProvider({"Task900_ToyA": fa, "Task901_ToyB": fb}, fold=0))
    torch.manual_seed(12345)
    np.random.seed(7)
    tr = get_trainer_class("rehearsal")("seg_outputs", "Task900_ToyA", plans=dict(PLANS), data_provider=prov, device=DEV, fold=0)
    tr.initialize(True, num_epochs=1)
    tr.num_batches_per_epoch, tr.num_val_batches_per_epoch = 3, 1
    sd0 = {k: v.detach().cpu().clone() for k, v in tr.network.state_dict().items()}
    losses = []
    orig = tr.run_iteration

    def it(gen, do_backprop=True, *a, **k):
        v = orig(gen, do_backprop, *a, **k)
        if do_backprop:
            losses.append(float(v))
        return v
    tr.run_iteration = it
    tr.run_training("Task900_ToyA")
    # ---- the three task-A iterations on the CPU oracle, with the batches the loader actually produced
    train_batches = [b for b in prov.seen if all(k in tr.dataset_tr for k in b["keys"])][:3]
    assert len(losses) == 3 and len(train_batches) == 3
    onet = OracleGenericUNet(1, 8, 3, 2)
    onet.load_state_dict(sd0)
    oopt = otrain.make_optimizer(onet)
    w = olosses.ds_loss_weights(2)
    exp = [otrain.run_iteration(onet, oopt, b["data"], b["target"], w)[0] for b in train_batches]
    print("task A from folders: hip", losses, "oracle", exp)
    for i, (g, e) in enumerate(zip(losses, exp)):
        assert abs(g - e) <= (1e-4 if i == 0 else 1e-3) * abs(e), (i, g, e)
    # ---- task B: rehearsal over the real folders (REH.py:105-164)
    tr.run_training("Task901_ToyB")
    fused = list(tr.dataset_tr.keys())
    nb = sum(k.startswith("Task901_ToyB") for k in fused)
    assert nb == 8 and len(tr.sampled["Task900_ToyA"]) == 2 and fused[nb:] == tr.sampled["Task900_ToyA"]
    assert all(os.path.isfile(tr.dataset_tr[k]["data_file"]) for k in fused)
    assert all(k.startswith("Task901_ToyB") for k in tr.dataset_val)
    assert all(np.isfinite(losses)) and len(losses) == 6
    # ---- validate(): both heads, whole validation cases, evaluator dictionary; pinned by the CPU oracle's tiled predictor
    out_dir = str(tmp_path / "val_out")
    res = tr.validate(do_mirroring=True, step_size=0.5, output_folder=out_dir, save_softmax=True)
    assert [r["task"] for r in res] == ["Task900_ToyA", "Task901_ToyB"]
    assert tr.already_trained_on["0"]["finished_validation_on"] == ["Task901_ToyB"]
    assert tr.mh_network.active_task == "Task901_ToyB" and tr.network.training
    for r, folder in zip(res, (fa, fb)):
        task = r["task"]
        assert len(r["cases"]) == 2 and all(k.startswith(task) for k in r["cases"])          # 5-fold split of 10 cases
        assert os.path.isfile(os.path.join(out_dir, "validation_raw" + task, "summary.json"))
        onet = OracleGenericUNet(1, 8, 3, 2)
        tr.network = tr.mh_network.assemble_model(task)
        onet.load_state_dict({k: v.detach().cpu() for k, v in tr.network.state_dict().items()})
        for case, masks in r["cases"].items():
            data = np.load(os.path.join(folder, case + ".npz"))["data"]
            oseg, _ = oinf.predict_3d_tiled(onet, data[:-1], PLANS["patch_size"], 0.5, True, (0, 1, 2), True)
            stored = np.load(os.path.join(out_dir, "validation_raw" + task, case + ".npz"))
            assert stored["seg"].shape == data.shape[1:] and stored["softmax"].shape == (3,) + data.shape[1:]
            agree = float((stored["seg"] == oseg).mean())
            exp_masks = oev.case_scores(oseg, data[-1], 2)
            print(task, case, "voxel agreement with the oracle predictor", agree, masks, exp_masks)
            assert agree >= 0.995
            assert masks == oev.case_scores(stored["seg"], data[-1], 2)       # the dictionary is the reference's arithmetic
            for m in masks:
                a, b = masks[m]["Dice"], exp_masks[m]["Dice"]
                assert (a is None) == (b is None) and (a is None or abs(a - b) <= 2e-2)
    tr.network = tr.mh_network.assemble_model("Task901_ToyB")


def test_head_logits_for_a_split_deeper_than_seg_outputs():
    """--split_at is arbitrary in the reference (run_training.py:103): with the split at ``tu`` a head holds the transposed
    convs and decoder blocks as well, and per-head evaluation is assemble_model + a complete forward (MHM.py:326-377,
    LWF.py:317-346).  head_logits(task) must equal the logits of a plain network carrying body + that head's tensors."""
    from lifelong_nnunet_amd.multihead import MultiHead_Module
    from lifelong_nnunet_amd.network import Generic_UNet
    torch.manual_seed(3)
    mh = MultiHead_Module(Generic_UNet, "tu", "A", None, 1, 8, 3, 2, device=DEV)
    assert not mh.head_is_seg_only() and any(n.startswith("tu.") for n, _ in mh.heads["A"].named_parameters())
    mh.add_new_task("B", use_init=False)
    with torch.no_grad():
        for p in mh.heads["B"].parameters():
            p.add_(0.05 * torch.randn_like(p))
    x = torch.randn((2, 1, 16, 32, 16))
    with pytest.raises(AssertionError):
        mh.head_weights("B")                      # the one-body-pass shortcut refuses deeper splits
    got = {t: mh.head_logits(t, x).cpu() for t in ("A", "B")}
    assert mh.active_task == "A"                  # restored
    for t in ("A", "B"):
        ref = OracleGenericUNet(1, 8, 3, 2)
        sd = {k: v.detach().cpu().clone() for k, v in mh.model.state_dict().items()}
        sd.update({k: v.detach().cpu().clone() for k, v in mh.heads[t].state_dict().items()})
        ref.load_state_dict(sd)
        ref.eval()
        with torch.no_grad():
            exp = ref(x)[0]
        err = float((got[t] - exp).abs().max() / exp.abs().max())
        print("split 'tu', head", t, "rel err vs oracle", err)
        assert err < 5e-3
    assert float((got["A"] - got["B"]).abs().max()) > 1e-3
    # seg_outputs split: the same call takes the one-body-pass path
    mh2 = MultiHead_Module(Generic_UNet, "seg_outputs", "A", None, 1, 8, 3, 2, device=DEV)
    assert mh2.head_is_seg_only() and tuple(mh2.head_logits("A", x).shape) == (2, 3, 16, 32, 16)
