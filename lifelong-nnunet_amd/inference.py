"""Sliding-window inference with Gaussian blending and test-time mirroring -- the step AFTER training that turns the
network into the reference's published metric.

Call sites in the reference: ``nnunet_ext/inference/predict.py:208-219`` and ``nnUNetTrainerMultiHead.validate``
(``MH.py:1052-1135``) call ``trainer.predict_preprocessed_data_return_seg_and_softmax(d, do_mirroring=...,
mirror_axes=..., use_sliding_window=True, step_size=0.5, use_gaussian=True, all_in_gpu=...)``; the tiling itself is
upstream nnU-Net v1 (``SegmentationNetwork.predict_3D`` -> ``_internal_predict_3D_3Dconv_tiled``), absent from
``/root/reference`` and restated here from its published algorithm:

  * the volume is zero-padded to at least the patch size (half of the difference below, the rest above);
  * per axis ``num_steps = ceil((image - patch) / (patch * step_size)) + 1`` tile origins spread evenly over
    ``[0, image - patch]`` (rounded);
  * importance map = ``scipy.ndimage.gaussian_filter`` of a unit impulse at ``patch // 2`` with ``sigma = patch / 8``,
    scaled to maximum 1, zeros replaced by the smallest non-zero value; used when there is more than one tile;
  * per tile: mean over the ``2^len(mirror_axes)`` mirrored passes of ``softmax(network(flip(tile)))`` flipped back,
    times the importance map, added into the aggregate; the map itself is added into the normaliser;
  * class probabilities = aggregate / normaliser (cropped back to the volume), segmentation = argmax.

The network forward is the HIP engine (batch 1, ``do_ds`` off); softmax, flip-back, weighting and accumulation are ONE
fused launch per mirrored pass (``lnn_softmax_accumulate``); normalisation + argmax one more (``lnn_softmax_finalize``).

Attribution: the algorithms restated in this file (function names kept so that callers read like upstream's) are those of
nnU-Net v1 (https://github.com/MIC-DKFZ/nnUNet, commit 77bc485, Apache License 2.0, Isensee et al., Nature Methods 2021);
no upstream source text is included.
"""
from functools import lru_cache
from typing import Sequence, Tuple

import numpy as np
import torch

from . import native as nat


def compute_steps_for_sliding_window(patch_size: Sequence[int], image_size: Sequence[int], step_size: float):
    assert all(i >= j for i, j in zip(image_size, patch_size)), "image size must be as large or larger than patch_size"
    assert 0 < step_size <= 1, "step_size must be larger than 0 and smaller or equal to 1"
    target = [i * step_size for i in patch_size]
    num_steps = [int(np.ceil((i - k) / j)) + 1 for i, j, k in zip(image_size, target, patch_size)]
    steps = []
    for dim in range(len(patch_size)):
        max_step_value = image_size[dim] - patch_size[dim]
        actual = max_step_value / (num_steps[dim] - 1) if num_steps[dim] > 1 else 99999999999
        steps.append([int(np.round(actual * i)) for i in range(num_steps[dim])])
    return steps


@lru_cache(maxsize=8)
def get_gaussian(patch_size: Tuple[int, ...], sigma_scale: float = 1.0 / 8) -> np.ndarray:
    from scipy.ndimage import gaussian_filter          # host-side constant of the method, computed once per patch size
    tmp = np.zeros(patch_size)
    tmp[tuple(i // 2 for i in patch_size)] = 1
    g = gaussian_filter(tmp, [i * sigma_scale for i in patch_size], 0, mode='constant', cval=0)
    g = (g / np.max(g) * 1).astype(np.float32)
    g[g == 0] = np.min(g[g != 0])
    return g


def pad_to_patch(data: torch.Tensor, patch_size: Sequence[int]):
    """(C, D, H, W) -> zero-padded to >= patch_size, plus the slicer that crops the result back."""
    shp = data.shape[1:]
    new = [max(s, p) for s, p in zip(shp, patch_size)]
    diff = [n - s for n, s in zip(new, shp)]
    below = [d // 2 for d in diff]
    above = [d // 2 + d % 2 for d in diff]
    if any(diff):
        pad = []
        for b, a in zip(reversed(below), reversed(above)):
            pad += [b, a]
        data = torch.nn.functional.pad(data, pad, mode="constant", value=0.0)
    slicer = tuple(slice(b, b + s) for b, s in zip(below, shp))
    return data, slicer


def predict_3D(network, x, do_mirroring: bool = True, mirror_axes: Tuple[int, ...] = (0, 1, 2), use_sliding_window: bool = True,
               step_size: float = 0.5, patch_size: Sequence[int] = None, use_gaussian: bool = True, verbose: bool = False):
    """``x``: (C, D, H, W) float array / tensor (a preprocessed case).  Returns ``(segmentation (D,H,W) int64 numpy,
    class_probabilities (K,D,H,W) float32 numpy)`` like upstream ``predict_3D``."""
    assert use_sliding_window, "only the tiled predictor is implemented (the reference always passes use_sliding_window=True)"
    assert patch_size is not None and len(patch_size) == 3
    if max(mirror_axes, default=0) > 2:
        raise ValueError("mirror axes. duh")
    dev = network.device_
    data = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x, dtype=torch.float32)
    assert data.dim() == 4, "x must be (c, x, y, z)"
    data, slicer = pad_to_patch(data, patch_size)
    data = data.to(dev)
    D, H, W = data.shape[1:]
    pd, ph, pw = patch_size
    steps = compute_steps_for_sliding_window(patch_size, (D, H, W), step_size)
    num_tiles = len(steps[0]) * len(steps[1]) * len(steps[2])
    gauss = None
    if use_gaussian and num_tiles > 1:
        gauss = torch.from_numpy(get_gaussian(tuple(patch_size))).to(dev)
    K = network.num_classes
    agg = torch.zeros((K, D, H, W), device=dev)
    nb = torch.zeros((D, H, W), device=dev)
    combos = [0]
    if do_mirroring:
        combos = [m for m in range(8) if all(((m >> (2 - a)) & 1) == 0 or a in mirror_axes for a in range(3))]
    weight = 1.0 / len(combos)
    was_ds, was_training = network.do_ds, network.training
    network.do_ds = False
    network.eval()
    try:
        with torch.no_grad():
            for z in steps[0]:
                for y in steps[1]:
                    for xx in steps[2]:
                        tile = data[None, :, z:z + pd, y:y + ph, xx:xx + pw]
                        for ci, m in enumerate(combos):
                            dims = [d + 2 for d in range(3) if (m >> (2 - d)) & 1]
                            inp = torch.flip(tile, dims) if dims else tile
                            logits = network(inp.contiguous())            # (1, K, pd, ph, pw) fp32, HIP engine
                            nat.call("lnn_softmax_accumulate", logits, gauss, agg, nb, K, pd, ph, pw, D, H, W, z, y, xx, m,
                                     weight, 1 if ci == 0 else 0)
            seg = torch.empty((D, H, W), dtype=torch.int32, device=dev)
            nat.call("lnn_softmax_finalize", agg, nb, K, D * H * W, seg)
    finally:
        network.do_ds = was_ds
        network.train(was_training)
    sl = (slice(None),) + slicer
    return seg[slicer].long().cpu().numpy(), agg[sl].cpu().numpy()


def dice_per_class(seg, label, num_classes):
    """evaluator2.py:91-105 / MH.py:1013-1019: Dice = 2TP / (2TP + FP + FN), IoU = TP / (TP + FP + FN) per foreground
    class of one subject (NaN when the class is absent from both)."""
    seg, label = np.asarray(seg), np.asarray(label)
    out = {}
    for c in range(1, num_classes):
        a, b = seg == c, label == c
        tp, fp, fn = float((a & b).sum()), float((a & ~b).sum()), float((~a & b).sum())
        den = 2 * tp + fp + fn
        out[c] = {"Dice": 2 * tp / den if den > 0 else float("nan"), "IoU": tp / (tp + fp + fn) if den > 0 else float("nan")}
    return out
