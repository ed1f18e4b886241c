"""Oracle training iteration, EWC after_train, LwF iteration value, rehearsal sampling (CPU, fp32).

Follows (reference paths relative to /root/reference):
  * iteration (fp32 branch):  multihead/nnUNetTrainerMultiHead.py:598-656  (clip 12 at :640)
  * optimiser:                multihead/nnUNetTrainerMultiHead.py:294-301 (SGD momentum .99 nesterov)
  * EWC iteration/after_train: ewc/nnUNetTrainerEWC.py:232-250, 252-310
  * LwF iteration:            lwf/nnUNetTrainerLWF.py:298-370 ; utilities/helpful_functions.py:207-266
  * rehearsal sampling:       rehearsal/nnUNetTrainerRehearsal.py:73,132

TEST INFRASTRUCTURE ONLY -- never imported by the product package.
"""
from __future__ import annotations

import random
from collections import OrderedDict

import torch

from . import losses


def make_optimizer(net, lr=1e-2, weight_decay=3e-5):
    params = [p for p in net.parameters() if p.requires_grad]
    return torch.optim.SGD(params, lr, weight_decay=weight_decay, momentum=0.99, nesterov=True)


def run_iteration(net, opt, data, target, weights, do_backprop=True, batch_dice=False, extra_loss=None,
                  clip=12.0):
    """One fp32 optimisation step; returns (loss value, outputs).  ``extra_loss`` is a callable
    returning a tensor added to the loss (EWC penalty, with autograd; LwF KL, detached)."""
    opt.zero_grad()
    out = net(data)
    l = losses.multiple_output_loss(out, target, weights, batch_dice)
    if extra_loss is not None:
        l = l + extra_loss()
    if do_backprop:
        l.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), clip)
        opt.step()
    return float(l.detach()), out


def ewc_after_train(net, opt, batches, weights, loss_extra=None, batch_dice=False):
    """ewc/nnUNetTrainerEWC.py:252-310: zero_grad before every batch, no step; Fisher is the squared
    gradient of the LAST batch only; theta* is a clone of the parameters."""
    net.train()
    opt.zero_grad()
    for data, target in batches:
        opt.zero_grad()
        out = net(data)
        l = losses.multiple_output_loss(out, target, weights, batch_dice)
        if loss_extra is not None:
            l = l + loss_extra()
        l.backward()
    fisher, params = OrderedDict(), OrderedDict()
    for name, p in net.named_parameters():
        fisher[name] = torch.tensor([1.0]) if p.grad is None else p.grad.data.clone().pow(2)
        params[name] = p.data.clone()
    return fisher, params


def lwf_loss_value(base_loss, pred_logits, target_logits, temperature=2.0):
    """deep_supervision.py:201-214 -- KL terms are added to the value; they carry no gradient because
    the predictions are detached (lwf/nnUNetTrainerLWF.py:343)."""
    l = base_loss
    for idx, t in enumerate(target_logits):
        l = l + losses.lwf_distillation(pred_logits[idx].detach(), t.detach(), temperature)
    return l


def rehearsal_sample(train_keys_per_prev_task, samples_in_perc=0.25, seed=3299):
    """rehearsal/nnUNetTrainerRehearsal.py:73,132 -- one ``random.seed(seed)`` then, per previous task in
    head order, ``random.sample(items, round(len * perc))`` over the (sorted) train keys."""
    random.seed(seed)
    picked = []
    for keys in train_keys_per_prev_task:
        keys = list(keys)
        picked.append(random.sample(keys, round(len(keys) * samples_in_perc)))
    random.seed()
    return picked
