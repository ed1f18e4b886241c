"""Per-case IoU / Dice exactly as the reference's evaluator computes them (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Follows nnunet_ext/evaluation/evaluator2.py:88-107: for c in 1 .. num_classes (foreground classes),
``tn, fp, fn, tp = sklearn.metrics.confusion_matrix((target == c).flatten(), (output == c).flatten(), labels=[False, True]).ravel()``,
``None`` scores when tp + fp + fn == 0, else IoU = tp / (tp + fp + fn), Dice = 2 tp / (2 tp + fp + fn).  Same call into
scikit-learn as the reference makes; the NIfTI reading around it (SimpleITK, evaluator2.py:80-86) is not part of the arithmetic."""
import numpy as np
import sklearn.metrics


def case_scores(output, target, num_classes):
    output, target = np.asarray(output).astype(int), np.asarray(target).astype(int)
    assert np.all(output.shape == target.shape)
    masks = dict()
    for c in range(1, num_classes + 1):
        tn, fp, fn, tp = sklearn.metrics.confusion_matrix((target == c).flatten(), (output == c).flatten(), labels=[False, True]).ravel()
        if tp + fp + fn == 0:
            iou = dice = None
        else:
            iou = tp / (tp + fp + fn)
            dice = 2 * tp / (2 * tp + fp + fn)
        masks['mask_' + str(c)] = {"IoU": iou, "Dice": dice}
    return masks
