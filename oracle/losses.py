"""Oracle losses: Dice+CE with deep supervision, EWC penalty, LwF distillation KL (CPU, fp32).

Follows (reference paths relative to /root/reference):
  * deep-supervision weights:   multihead/nnUNetTrainerMultiHead.py:1373-1383
  * loss construction:          multihead/nnUNetTrainerMultiHead.py:1385-1386, ewc/nnUNetTrainerEWC.py:131-140,
                                lwf/nnUNetTrainerLWF.py:103-108
  * EWC penalty:                nnunet_ext/training/loss_functions/deep_supervision.py:58-83
  * LwF distillation:           nnunet_ext/training/loss_functions/deep_supervision.py:185-214
  * upstream DC_and_CE_loss / SoftDiceLoss / MultipleOutputLoss2 (nnunet @77bc485)

TEST INFRASTRUCTURE ONLY -- never imported by the product package.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def ds_loss_weights(num_pool: int) -> np.ndarray:
    """multihead/nnUNetTrainerMultiHead.py:1377-1383."""
    w = np.array([1 / (2 ** i) for i in range(num_pool)])
    mask = np.array([True] + [i < num_pool - 1 for i in range(1, num_pool)])
    w[~mask] = 0
    return w / w.sum()


def soft_dice_loss(logits, target, batch_dice=False, smooth=1e-5, do_bg=False):
    """upstream SoftDiceLoss(apply_nonlin=softmax_helper, batch_dice, do_bg, smooth)."""
    p = torch.softmax(logits, 1)
    onehot = torch.zeros_like(p)
    onehot.scatter_(1, target.long(), 1)
    axes = ([0] if batch_dice else []) + list(range(2, logits.dim()))
    tp = (p * onehot).sum(axes)
    fp = (p * (1 - onehot)).sum(axes)
    fn = ((1 - p) * onehot).sum(axes)
    dc = (2 * tp + smooth) / (2 * tp + fp + fn + smooth + 1e-8)
    if not do_bg:
        dc = dc[1:] if batch_dice else dc[:, 1:]
    return -dc.mean()


def dc_and_ce_loss(logits, target, batch_dice=False):
    """upstream DC_and_CE_loss({'batch_dice', 'smooth': 1e-5, 'do_bg': False}, {}) -> ce + dice."""
    ce = F.cross_entropy(logits, target[:, 0].long())
    return ce + soft_dice_loss(logits, target, batch_dice)


def multiple_output_loss(outputs, targets, weights, batch_dice=False):
    """upstream MultipleOutputLoss2.forward: zero-weight levels are skipped, not multiplied."""
    l = weights[0] * dc_and_ce_loss(outputs[0], targets[0], batch_dice)
    for i in range(1, len(outputs)):
        if weights[i] != 0:
            l = l + weights[i] * dc_and_ce_loss(outputs[i], targets[i], batch_dice)
    return l


def ewc_penalty(named_params, fisher, params, ewc_lambda=0.4, first_task_only=True):
    """deep_supervision.py:62-80.  ``first_task_only`` reproduces the generator-exhaustion behaviour
    of the base EWC trainer (ewc/nnUNetTrainerEWC.py:140,247 hands over a *generator*, so only the
    first task in ``fisher`` ever sees parameters); False is the list behaviour of the ewc_unet
    variants."""
    named_params = list(named_params)
    pen = None
    for ti, task in enumerate(fisher.keys()):
        if first_task_only and ti > 0:
            break
        for name, p in named_params:
            t = ewc_lambda / 2 * (fisher[task][name] * (p - params[task][name]).pow(2)).sum()
            pen = t if pen is None else pen + t
    return pen if pen is not None else torch.zeros(())


def rw_penalty(named_params, fisher, params, importance, rw_lambda=0.4, first_task_only=True):
    """deep_supervision.py:110-135 (MultipleOutputLossRW.forward): for every task but the one being trained
    (``tasks = list(fisher)[:-1]``, :106) ``lambda * sum((F + S) * (theta - theta*)^2)`` -- no 1/2, unlike EWC.
    ``first_task_only``: the RW trainer hands the loss ``network.named_parameters()`` (rw/nnUNetTrainerRW.py:122-125,
    a generator) and never refreshes it, so the nested loop only ever sees parameters for the first task of the FIRST
    forward with previous tasks; afterwards the penalty is silently zero.  Callers model "afterwards" by passing an
    empty ``named_params``."""
    named_params = list(named_params)
    pen = None
    for ti, task in enumerate(list(fisher.keys())[:-1]):
        if first_task_only and ti > 0:
            break
        for name, p in named_params:
            t = rw_lambda * ((fisher[task][name] + importance[task][name]) * (p - params[task][name]).pow(2)).sum()
            pen = t if pen is None else pen + t
    return pen if pen is not None else torch.zeros(())


def unbiased_kd(inputs, targets, alpha=1.0):
    """knowledge_distillation.py:11-32 (UnbiasedKnowledgeDistillationLoss, reduction 'mean', no mask), restated for any
    class counts: ``new_cl`` = K_new - K_old, or K when they are equal (the reference's own quirk at :12)."""
    k_in, k_t = inputs.shape[1], targets.shape[1]
    new_cl = k_in - k_t if k_in != k_t else k_in
    t = targets * alpha
    den = torch.logsumexp(inputs, dim=1)
    no_bkg = (inputs[:, 1:-new_cl] if (k_in - new_cl) > 1 else inputs[:, 1:]) - den.unsqueeze(1)
    idx = torch.tensor([0] + list(range(k_t, k_in)))
    bkg = torch.logsumexp(inputs.index_select(1, idx), dim=1) - den
    lab = torch.softmax(t, dim=1)
    loss = (lab[:, 0] * bkg + (lab[:, 1:] * no_bkg).sum(dim=1)) / k_t
    return -loss.mean()


def mib_loss(outputs, old_outputs, targets, weights, alpha=1.0, lkd=10.0, ignore_index=255):
    """deep_supervision.py:401-416: sum_i w_i CE(x_i, y_i) (zero weights skipped) + sum_i w_i * lkd * UKD(x_i, x_o_i)."""
    l = None
    for i, (x, y) in enumerate(zip(outputs, targets)):
        if i > 0 and weights[i] == 0:
            continue
        t = weights[i] * torch.nn.functional.cross_entropy(x, y[:, 0].long(), ignore_index=ignore_index)
        l = t if l is None else l + t
    for i, (x, xo) in enumerate(zip(outputs, old_outputs)):
        l = l + weights[i] * lkd * unbiased_kd(x, xo, alpha)
    return l


def lwf_distillation(pred_logits, teacher_logits, temperature=2.0):
    """deep_supervision.py:194-196: batchmean KL between log-softmaxes at temperature T (no T^2)."""
    return F.kl_div(F.log_softmax(pred_logits.float() / temperature, dim=1),
                    F.log_softmax(teacher_logits.float() / temperature, dim=1),
                    reduction='batchmean', log_target=True)


def online_dice_counts(logits, target):
    """multihead/nnUNetTrainerMultiHead.py:938-951: per-sample hard TP/FP/FN for the foreground classes."""
    num_classes = logits.shape[1]
    seg = torch.softmax(logits, 1).argmax(1)
    tgt = target[:, 0]
    axes = tuple(range(1, tgt.dim()))
    tp = torch.zeros((tgt.shape[0], num_classes - 1))
    fp = torch.zeros_like(tp)
    fn = torch.zeros_like(tp)
    for c in range(1, num_classes):
        tp[:, c - 1] = ((seg == c).float() * (tgt == c).float()).sum(axes)
        fp[:, c - 1] = ((seg == c).float() * (tgt != c).float()).sum(axes)
        fn[:, c - 1] = ((seg != c).float() * (tgt == c).float()).sum(axes)
    return tp, fp, fn


def dice_from_counts(tp, fp, fn):
    """multihead/nnUNetTrainerMultiHead.py:1017-1019."""
    return 2 * tp / (2 * tp + fp + fn), tp / (tp + fp + fn)
