"""Deterministic synthetic patches shaped like the reference's data dict.

The reference's iteration consumes ``{'data': (B,C,D,H,W) f32, 'target': [ (B,1,D/2^i,H/2^i,W/2^i) ]_i,
'keys'}`` (multihead/nnUNetTrainerMultiHead.py:606-608).  There is no dataset in this build, so the
patches are synthetic: z-scored Gaussian intensities and nested smooth blob labels {0,1,2}, generated
on the CPU from a seed so the oracle and the GPU path see bit-identical inputs.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _smooth_field(shape, gen, passes=3, k=5):
    f = torch.randn((1, 1) + tuple(shape), generator=gen)
    for _ in range(passes):
        f = F.avg_pool3d(F.pad(f, (k // 2,) * 6, mode='replicate'), k, 1)
    f = f - f.mean()
    return f / (f.std() + 1e-8)


def ds_strides(num_pool, pool_op_kernel_sizes=None):
    """Cumulative per-axis strides of the deep-supervision levels 0 .. num_pool - 1: upstream's ``deep_supervision_scales``
    (``1 / cumprod(net_num_pool_op_kernel_sizes)``, the lowest resolution dropped); 2^i per axis for the isotropic plans."""
    pools = [(2, 2, 2)] * num_pool if pool_op_kernel_sizes is None else [tuple(int(v) for v in q) for q in pool_op_kernel_sizes]
    out, cur = [], (1, 1, 1)
    for i in range(num_pool):
        out.append(cur)
        cur = tuple(c * q for c, q in zip(cur, pools[i]))
    return out


def make_patch_batch(batch, patch, num_pool, in_channels=1, num_labels=3, seed=12345, blob_scale=1.0, pool_op_kernel_sizes=None):
    """Returns ``(data, targets)``: data (B,C,D,H,W) f32, targets list of ``num_pool`` tensors
    (B,1,D/s_i,...) f32 holding integer labels (nearest-neighbour subsampled at the cumulative pooling strides s_i)."""
    gen = torch.Generator().manual_seed(int(seed))
    data = torch.randn((batch, in_channels) + tuple(patch), generator=gen)
    full = torch.zeros((batch, 1) + tuple(patch))
    for b in range(batch):
        f = _smooth_field(patch, gen, k=max(3, int(5 * blob_scale) | 1))[0, 0]
        lab = torch.zeros_like(f)
        # nested blobs: class 1 where the field is high, class 2 in its core
        # (more than three labels: further nested shells; the thresholds of the first two are unchanged)
        levels = [0.70, 0.90] + [0.90 + 0.09 * (j + 1) / (num_labels - 3) for j in range(max(0, num_labels - 3))]
        qs = torch.quantile(f.flatten()[:: max(1, f.numel() // 65536)], torch.tensor(levels))
        lab[f > qs[0]] = 1
        for c in range(2, num_labels):
            lab[f > qs[c - 1]] = c
        full[b, 0] = lab
        # make the image weakly informative about the label so training has signal
        data[b] += 0.75 * lab[None]
    targets = [full[:, :, ::sz, ::sy, ::sx].contiguous() for sz, sy, sx in ds_strides(num_pool, pool_op_kernel_sizes)]
    return data, targets


class SyntheticPatchGenerator:
    """Infinite iterator of data dicts; ``period`` distinct batches are cycled deterministically
    (the reference pairs batch ``i mod 250`` with stored teacher logits, lwf/nnUNetTrainerLWF.py:349)."""

    def __init__(self, batch, patch, num_pool, in_channels=1, num_labels=3, seed=12345, period=4,
                 blob_scale=1.0, key_prefix="case", pool_op_kernel_sizes=None):
        self.batches = [make_patch_batch(batch, patch, num_pool, in_channels, num_labels, seed + 1000 * i,
                                         blob_scale, pool_op_kernel_sizes) for i in range(period)]
        self.keys = [[f"{key_prefix}_{i:03d}_{b}" for b in range(batch)] for i in range(period)]
        self.i = 0

    def __iter__(self):
        return self

    def __next__(self):
        d, t = self.batches[self.i % len(self.batches)]
        k = self.keys[self.i % len(self.batches)]
        self.i += 1
        return {'data': d, 'target': t, 'keys': k}
