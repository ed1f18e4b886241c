"""Oracle restatement of the PLOP / POD path (CPU, fp32, plain torch).  TEST INFRASTRUCTURE ONLY -- never imported by
the product package.  Pinned by tests/golden/plop_reference.{npz,json}, which oracle/make_goldens_plop.py produced by
EXECUTING the reference's ``local_POD``, ``MultipleOutputLossPLOP`` / ``MultipleOutputLossPOD`` and the
``nnUNetTrainerPLOP`` / ``nnUNetTrainerPOD`` methods.

Follows (paths relative to /root/reference/nnunet_ext):
  * ``entropy``                training/loss_functions/crossentropy.py:6-16
  * ``local_pod``              training/loss_functions/embeddings.py:3-41
  * ``pseudo_label_loss``      training/loss_functions/deep_supervision.py:292-334
  * ``plop_loss`` / ``pod_loss``   deep_supervision.py:248-290 / :362-381
  * ``thresholds``             training/network_training/plop/nnUNetTrainerPLOP.py:114-172
  * ``Taps``                   plop/nnUNetTrainerPLOP.py:335-358 (forward hooks on every conv / transposed conv)

Behaviour of the reference AS ITS CODE RUNS that is reproduced here (each is visible in the fixtures):
  * the hooks store ``output.detach()`` for BOTH networks, so the POD term has no gradient;
  * scale 0 adds no window (``range(0, W - w, w)`` is empty for w == W) and the last window of every other scale is
    skipped; rows of the window are cut with the WIDTH step and vice versa, so non-square last-two dims fail in
    ``torch.cat`` -- only H == W works;
  * ``dist_loss /= num_layers`` sits inside the layer loop (deep_supervision.py:276): layer i of L is weighted L**-(L-i+1);
  * the (B,D,H,W) masks are summed over dims (1,2) only, so the adaptive factor is the mean of a (B,W) table of
    per-column ratios and multiplies the two scalar CE terms;
  * ``extract_max_entropy_and_thresholds`` compares a LIST of label tensors with 0 (``labels == 0`` -> False), the
    histograms stay empty and every threshold is the 0.001 floor; it still draws ``num_batches_per_epoch`` batches from
    the generator of the task trained BEFORE (the new task's loaders are created later);
  * from the third task on the old model is a deepcopy of a network that already carries the current-model hooks, so its
    forward overwrites ``interm_results`` with its own activations: the POD term is exactly 0.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn


def entropy(probs):
    """-1/log(K+1e-8) * mean_k p_k log(p_k + 1e-8)"""
    K = probs.shape[1]
    return -(probs * torch.log(probs + 1e-8)).mean(1) / math.log(K + 1e-8)


def local_pod(h, h_old, scales):
    """Windowed formulation: for scale s >= 1 the windows are the first P = len(range(0, S - win, win)) tiles of side
    win = S >> s in both of the last two dims; per tile the row means and the column means of (h - h_old)."""
    assert h.shape == h_old.shape
    S = h.shape[-1]
    if h.shape[-2] != S:
        raise RuntimeError("local POD needs equal last two dims (the reference's torch.cat fails otherwise)")
    d = (h.double() - h_old.double())
    sq_w = torch.zeros(h.shape[:-2], dtype=torch.float64)
    sq_h = torch.zeros(h.shape[:-2], dtype=torch.float64)
    for s in range(scales):
        win = int(S / 2 ** s)
        assert win > 0, "too many scales"
        P = len(range(0, S - win, win))
        if P == 0:
            continue
        t = d[..., :P * win, :P * win].reshape(*h.shape[:-2], P, win, P, win)
        sq_w += (t.mean(-1) ** 2).sum((-1, -2, -3))       # width-pooled: one value per (tile row, row in tile, tile col)
        sq_h += (t.mean(-3) ** 2).sum((-1, -2, -3))       # height-pooled
    return float((sq_w.sqrt().sum() + sq_h.sqrt().sum()) / (2 * sq_w.numel()))


def pseudo_labels(x_o, y, thr, max_entropy):
    """y: (B,D,H,W) float labels.  Returns (labels_not_pseudo, labels_pseudo, factor)."""
    probs = torch.softmax(x_o, 1)
    pl = probs.argmax(1)
    valid = (entropy(probs) / max_entropy) < thr[pl]
    bg = y == 0
    m = valid & bg
    lab = torch.where(m, torch.full_like(y, 255.), y)
    pse = torch.where(m, pl.float(), torch.full_like(y, 255.))
    num = m.float().sum((1, 2))
    den = bg.float().sum((1, 2))
    return lab, pse, (num / den).mean()


def pseudo_label_loss(x, x_o, y, thr, max_entropy):
    """mean_{b,w}(num/den) * (CE(x, pseudo labels) + CE(x, labels without the pseudo voxels)), ignore_index 255."""
    y = y.squeeze()
    assert y.dim() == x.dim() - 1, "batch size 1 is squeezed away in the reference too (and then fails in CE)"
    lab, pse, factor = pseudo_labels(x_o, y, thr, max_entropy)
    ce = lambda t: F.cross_entropy(x, t.long(), ignore_index=255)
    return factor * (ce(pse) + ce(lab))


def dist_term(interm, old_interm, pod_lambda, scales):
    L = len(old_interm)
    dist = 0.
    for name, h_old in old_interm.items():
        dist = (dist + pod_lambda * local_pod(interm[name], h_old, scales)) / L
    return dist


def plop_loss(x, x_o, y, weights, interm, old_interm, thresholds, max_entropy, pod_lambda, scales):
    loss = weights[0] * pseudo_label_loss(x[0], x_o[0], y[0], thresholds[0], max_entropy)
    for i in range(1, len(x)):
        if weights[i] != 0:
            loss = loss + weights[i] * pseudo_label_loss(x[i], x_o[i], y[i], thresholds[i], max_entropy)
    return loss + dist_term(interm, old_interm, pod_lambda, scales)


def pod_loss(base_value, interm, old_interm, pod_lambda, scales):
    return base_value + dist_term(interm, old_interm, pod_lambda, scales)


def thresholds(num_classes, num_levels, base_threshold=0.001):
    """What extract_max_entropy_and_thresholds leaves behind (see the module docstring): the floor everywhere."""
    return {i: torch.full((num_classes,), base_threshold) for i in range(num_levels)}, math.log(num_classes)


class Taps:
    """Detached outputs of every Conv3d / ConvTranspose3d in execution order, as the reference's hooks record them."""

    def __init__(self, net):
        self.out = OrderedDict()
        self.handles = []
        for name, mod in net.named_modules():
            if isinstance(mod, (nn.Conv3d, nn.ConvTranspose3d)):
                self.handles.append(mod.register_forward_hook(self._hook(name)))

    def _hook(self, name):
        def hook(mod, inp, out):
            self.out[name] = out.detach()
        return hook

    def take(self):
        o, self.out = self.out, OrderedDict()
        return o

    def remove(self):
        for h in self.handles:
            h.remove()


def plop_iteration(net, net_old, taps, taps_old, opt, data, target, weights, thr, max_entropy, pod_lambda, scales,
                   pod_only=False, base_loss=None, alias_old=False, clip=12.0):
    """One PLOP / POD training iteration (PLOP.py:217-333): forward of both models on the same batch, loss, backward, clip,
    SGD step.  ``alias_old``: third-task behaviour (interm_results overwritten by the old model's activations)."""
    opt.zero_grad()
    out = net(data)
    interm = taps.take()
    with torch.no_grad():
        out_o = net_old(data)
    old = taps_old.take()
    if alias_old:
        interm = old
    if pod_only:
        loss = base_loss(out, target) + dist_term(interm, old, pod_lambda, scales)
    else:
        loss = plop_loss(out, out_o, target, weights, interm, old, thr, max_entropy, pod_lambda, scales)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(net.parameters(), clip)
    opt.step()
    return float(loss.detach())
