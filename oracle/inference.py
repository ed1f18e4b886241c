"""CPU restatement of the tiled predictor (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Follows upstream nnU-Net v1 ``SegmentationNetwork._internal_predict_3D_3Dconv_tiled`` /
``_internal_maybe_mirror_and_pred_3D`` / ``_compute_steps_for_sliding_window`` / ``_get_gaussian`` (nnunet @ 77bc485,
``network_architecture/neural_network.py``; not under /root/reference -- reached from the reference at
``nnunet_ext/inference/predict.py:208-219`` and ``MH.py:1052-1135``).  PARITY UNPINNED: the reference holds no
numeric fixture for this step; the restatement is written independently of lifelong-nnunet_amd/inference.py (explicit
loops and array slicing instead of the fused kernel) and pinned by hand-computed step lists / map properties in
tests/test_oracle_golden.py."""
import numpy as np
import torch
from scipy.ndimage import gaussian_filter


def steps_for_sliding_window(patch_size, image_size, step_size):
    out = []
    for p, i in zip(patch_size, image_size):
        n = int(np.ceil((i - p) / (p * step_size))) + 1
        span = i - p
        if n > 1:
            out.append([int(np.round(span / (n - 1) * k)) for k in range(n)])
        else:
            out.append([0])
    return out


def gaussian_map(patch_size, sigma_scale=1.0 / 8):
    tmp = np.zeros(patch_size)
    tmp[tuple(i // 2 for i in patch_size)] = 1
    g = gaussian_filter(tmp, [i * sigma_scale for i in patch_size], 0, mode='constant', cval=0)
    g = (g / g.max()).astype(np.float32)
    g[g == 0] = g[g != 0].min()
    return g


def predict_3d_tiled(net, x, patch_size, step_size=0.5, do_mirroring=True, mirror_axes=(0, 1, 2), use_gaussian=True):
    """net: OracleGenericUNet (returns a tuple when do_ds, full resolution first); x: (C, D, H, W) float32 numpy."""
    x = np.asarray(x, dtype=np.float32)
    shp = x.shape[1:]
    new = [max(s, p) for s, p in zip(shp, patch_size)]
    below = [(n - s) // 2 for n, s in zip(new, shp)]
    padded = np.zeros((x.shape[0],) + tuple(new), dtype=np.float32)
    padded[(slice(None),) + tuple(slice(b, b + s) for b, s in zip(below, shp))] = x
    steps = steps_for_sliding_window(patch_size, new, step_size)
    n_tiles = len(steps[0]) * len(steps[1]) * len(steps[2])
    g = gaussian_map(tuple(patch_size)) if (use_gaussian and n_tiles > 1) else np.ones(patch_size, dtype=np.float32)
    K = None
    agg = nb = None
    net.eval()
    with torch.no_grad():
        for z in steps[0]:
            for y in steps[1]:
                for xx in steps[2]:
                    tile = torch.from_numpy(padded[None, :, z:z + patch_size[0], y:y + patch_size[1], xx:xx + patch_size[2]].copy())
                    res = None
                    flips = [()]
                    if do_mirroring:
                        flips = [tuple(a for a in range(3) if (m >> a) & 1) for m in range(8)]
                        flips = [f for f in flips if all(a in mirror_axes for a in f)]
                    for f in flips:
                        dims = [a + 2 for a in f]
                        out = net(torch.flip(tile, dims) if dims else tile)
                        out = out[0] if isinstance(out, (tuple, list)) else out
                        p = torch.softmax(out, 1)
                        p = torch.flip(p, dims) if dims else p
                        res = p / len(flips) if res is None else res + p / len(flips)
                    res = res[0].numpy() * g[None]
                    if agg is None:
                        K = res.shape[0]
                        agg = np.zeros((K,) + tuple(new), dtype=np.float32)
                        nb = np.zeros(tuple(new), dtype=np.float32)
                    agg[:, z:z + patch_size[0], y:y + patch_size[1], xx:xx + patch_size[2]] += res
                    nb[z:z + patch_size[0], y:y + patch_size[1], xx:xx + patch_size[2]] += g
    crop = tuple(slice(b, b + s) for b, s in zip(below, shp))
    probs = (agg / nb[None])[(slice(None),) + crop]
    return probs.argmax(0), probs
