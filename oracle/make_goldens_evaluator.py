"""Generates tests/golden/evaluator_reference.{npz,json}: per-case IoU / Dice as the REFERENCE's evaluator computes them.

``nnunet_ext/evaluation/evaluator2.py:60-109`` (``compute_scores_and_build_dict``) is imported from /root/reference and EXECUTED on
seeded label volumes: it reads ``plans.pkl`` (``num_classes`` = number of FOREGROUND classes), ``splits_final.pkl``, and for every
case of the fold one predicted and one ground-truth volume, and returns ``{case: {'mask_c': {'IoU', 'Dice'}}}`` with ``None`` where a
class is absent from both.  What is stood in for: the file readers it calls -- ``load_pickle / join / isfile`` of
``batchgenerators.utilities.file_and_folder_operations`` (pickle.load, os.path.join, os.path.isfile: oracle/ref_shim.py) and
``sitk.GetArrayFromImage(sitk.ReadImage(path))`` (the volumes are .npy files next to empty ``.nii.gz`` names: SimpleITK is not
installed and the NIfTI container is not part of the arithmetic) -- and the two directory constants it imports.  The scores come from
the reference's own lines, including its call into scikit-learn.

    python -m oracle.make_goldens_evaluator        (in the build container; /root/reference is not on the GPU box)

Only DATA is written (npz / json): no reference source or bytecode is copied."""
from __future__ import annotations

import json
import os
import pickle
import tempfile
import types

import numpy as np

from . import ref_shim
from .make_goldens import OUT


def main():
    ref_shim.install()          # join / isfile / load_pickle of batchgenerators' file utilities are its concrete stand-ins
    import nnunet_ext.evaluation.evaluator2 as ev
    ev.sitk = types.SimpleNamespace(ReadImage=lambda p: p, GetArrayFromImage=lambda p: np.load(p + ".npy"))
    rng = np.random.default_rng(77)
    num_classes = 3                      # foreground classes 1..3 (plans['num_classes'] as nnU-Net stores it)
    cases = {"case_a": (12, 10, 14), "case_b": (9, 16, 8), "case_c": (6, 6, 6), "case_d": (8, 8, 8), "case_t": (5, 7, 9)}
    arrays, vols = {}, {}
    for name, shape in cases.items():
        tgt = rng.integers(0, num_classes + 1, size=shape)
        out = np.where(rng.random(shape) < 0.7, tgt, rng.integers(0, num_classes + 1, size=shape))
        if name == "case_b":             # class 3 absent from both volumes -> None scores
            tgt[tgt == 3] = 0; out[out == 3] = 0
        if name == "case_c":             # class 2 only predicted (false positives only -> 0.0, not None)
            tgt[tgt == 2] = 1
        if name == "case_d":             # perfect prediction
            out = tgt.copy()
        vols[name] = (out.astype(np.int16), tgt.astype(np.uint8))
        arrays[f"out::{name}"], arrays[f"tgt::{name}"] = vols[name]
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        inf, pre, raw = os.path.join(tmp, "inference"), os.path.join(tmp, "pre"), os.path.join(tmp, "raw")
        task = "Task099_Toy"
        gt = os.path.join(raw, "nnUNet_raw_data", task, "labelsTr")
        for d in (inf, os.path.join(pre, task), gt):
            os.makedirs(d)
        pickle.dump({"num_classes": num_classes}, open(os.path.join(inf, "plans.pkl"), "wb"))
        splits = [{"train": np.array(["case_t"]), "val": np.array(["case_a", "case_b", "case_c", "case_d"])}]
        pickle.dump(splits, open(os.path.join(pre, task, "splits_final.pkl"), "wb"))
        for name, (o, t) in vols.items():
            for folder, arr in ((inf, o), (gt, t)):
                open(os.path.join(folder, name + ".nii.gz"), "wb").close()
                np.save(os.path.join(folder, name + ".nii.gz.npy"), arr)
        ev.preprocessing_output_dir = pre
        os.environ["nnUNet_raw_data_base"] = raw
        for inc in (False, True):
            d = ev.compute_scores_and_build_dict(task, inf, 0, inc)
            res["include_training_data" if inc else "validation_only"] = {
                str(c): {m: {k: (None if v is None else float(v)) for k, v in sc.items()} for m, sc in masks.items()} for c, masks in d.items()}
    np.savez_compressed(os.path.join(OUT, "evaluator_reference.npz"), **arrays)
    json.dump({"num_classes": num_classes, "fold": 0, "splits": {"train": ["case_t"], "val": ["case_a", "case_b", "case_c", "case_d"]},
               "results": res}, open(os.path.join(OUT, "evaluator_reference.json"), "w"), indent=1)
    print("wrote evaluator_reference.*:", {k: list(v) for k, v in res.items()})
    print(json.dumps(res["validation_only"]["case_b"]), json.dumps(res["validation_only"]["case_c"]))


if __name__ == "__main__":
    main()
