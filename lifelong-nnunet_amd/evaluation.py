"""Per-subject evaluation summary of full-volume predictions -- the dictionary the reference's evaluator builds.

Mirrors ``nnunet_ext/evaluation/evaluator2.py:60-109`` (``compute_scores_and_build_dict``): for every evaluated case and
every foreground class ``c = 1 .. num_classes`` (``plans['num_classes']`` counts the foreground classes only) the
confusion counts of ``output == c`` against ``target == c`` give

    IoU  = TP / (TP + FP + FN),     Dice = 2 TP / (2 TP + FP + FN),

stored as ``cases[case]['mask_<c>'] = {'IoU': .., 'Dice': ..}``; both are ``None`` when TP + FP + FN == 0 (the ground truth
holds only background and so does the prediction: evaluator2.py:96-101).  The reference reads the two volumes from NIfTI
files (NIfTI I/O is out of scope); here they arrive as arrays -- e.g. from
``nnUNetTrainerMultiHead.validate``.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Mapping, Tuple

import numpy as np


def case_scores(output, target, num_classes: int) -> Dict[str, Dict[str, float]]:
    """evaluator2.py:88-107 for one case.  ``num_classes`` = number of FOREGROUND classes (labels 1 .. num_classes)."""
    output = np.asarray(output).astype(int)
    target = np.asarray(target).astype(int)
    assert output.shape == target.shape, f"prediction {output.shape} and ground truth {target.shape} differ in shape"
    masks = OrderedDict()
    for c in range(1, num_classes + 1):
        o, t = output == c, target == c
        tp = int(np.count_nonzero(o & t))
        fp = int(np.count_nonzero(o & ~t))
        fn = int(np.count_nonzero(~o & t))
        if tp + fp + fn == 0:
            iou = dice = None
        else:
            iou = tp / (tp + fp + fn)
            dice = 2 * tp / (2 * tp + fp + fn)
        masks['mask_' + str(c)] = {"IoU": iou, "Dice": dice}
    return masks


def compute_scores_and_build_dict(cases: Mapping[str, Tuple[np.ndarray, np.ndarray]], num_classes: int):
    """``cases``: case identifier -> (predicted segmentation, ground-truth segmentation).  Returns the reference's
    ``cases_dict`` (evaluator2.py:72-108)."""
    return OrderedDict((case, case_scores(out, tgt, num_classes)) for case, (out, tgt) in cases.items())


def summarize(cases_dict) -> Dict[str, Dict[str, float]]:
    """Mean / std over the cases of every mask's IoU and Dice, ``None`` entries left out (what the reference's
    ``summarized_val_metrics`` tables report per mask); masks without any scored case map to ``None``."""
    out = OrderedDict()
    masks = sorted({m for c in cases_dict.values() for m in c}, key=lambda m: int(m.split('_')[1]))
    for m in masks:
        out[m] = {}
        for metric in ("IoU", "Dice"):
            vals = [c[m][metric] for c in cases_dict.values() if m in c and c[m][metric] is not None]
            out[m][metric] = {"mean": float(np.mean(vals)), "std": float(np.std(vals)), "n": len(vals)} if vals else None
    return out
