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


RW_EPSILON = 1e-8          # rw/nnUNetTrainerRW.py:17


def rw_new_task_state(net):
    """rw/nnUNetTrainerRW.py:163-169: zero Fisher / score per trainable parameter; prev_param None, count 0 (:179)."""
    fisher = OrderedDict((n, torch.zeros_like(p)) for n, p in net.named_parameters() if p.requires_grad)
    scores = OrderedDict((n, torch.zeros_like(p)) for n, p in net.named_parameters() if p.requires_grad)
    return {"fisher": fisher, "scores": scores, "prev_param": None, "count": 0}


def rw_update_f_s(net, st, alpha=0.9, fisher_update_after=10):
    """rw/nnUNetTrainerRW.py:231-265, called after every run_iteration (``param.grad`` is therefore the clipped
    gradient the optimiser step just used).  Every ``fisher_update_after`` iterations:
      score += max(0, g * (prev - theta) / (0.5 * F * (theta - prev)^2 + eps))      (only once a prev exists)
      prev   = theta
      F      = alpha * g^2 + (1 - alpha) * F
    for the parameters that have a gradient."""
    if st["count"] % fisher_update_after == 0:
        if st["prev_param"] is not None:
            for name, p in net.named_parameters():
                if p.grad is not None:
                    delta = p.grad.detach() * (st["prev_param"][name] - p.detach())
                    den = 0.5 * st["fisher"][name] * (p.detach() - st["prev_param"][name]).pow(2) + RW_EPSILON
                    sc = delta / den
                    sc[sc < 0] = 0
                    st["scores"][name] += sc
        st["prev_param"] = {k: v.detach().clone() for k, v in net.named_parameters() if v.grad is not None}
        for name, p in net.named_parameters():
            if p.grad is not None:
                st["fisher"][name] = alpha * p.grad.data.clone().pow(2) + (1 - alpha) * st["fisher"][name]
    st["count"] += 1


def rw_finish_task(net, st, n_finished, prev_scores=None):
    """rw/nnUNetTrainerRW.py:174-205, after the epochs of a task (``n_finished`` = len(finished_training_on), which
    already contains the task).  Quirks kept: the Fisher is normalised with the extrema of the per-tensor maxima of
    the SCORES (:184-188 reuse ``self.scores``); the scores are rescaled to [0, 2] for the first task (:193-197), left
    untouched for the second, and for >= 3 tasks averaged with ``scores[finished[-1]]`` which is the task itself
    (:198-204)."""
    params = OrderedDict((n, p.data.clone()) for n, p in net.named_parameters())
    values = [torch.max(v) for v in st["scores"].values()]
    minim, maxim = min(values), max(values)
    fisher = OrderedDict((k, (v - minim) / (maxim - minim + RW_EPSILON)) for k, v in st["fisher"].items())
    scores = OrderedDict((k, v.clone()) for k, v in st["scores"].items())
    if n_finished == 1:
        for k, v in st["scores"].items():
            scores[k] = 2 * ((v - minim) / (maxim - minim + RW_EPSILON))
    elif n_finished > 2:
        prev = {k: v.clone() for k, v in st["scores"].items()}
        for k, v in st["scores"].items():
            scores[k] = 0.5 * (prev[k] + (v - minim) / (maxim - minim + RW_EPSILON))
    return fisher, params, scores


def per_subject_dice(tp_batches, fp_batches, fn_batches, names_per_batch):
    """multihead/nnUNetTrainerMultiHead.py:963-1049: hard TP / FP / FN of all samples carrying the same subject name are
    summed, then IoU = TP/(TP+FP+FN) and Dice = 2TP/(2TP+FP+FN) per subject and foreground class (0/0 -> NaN)."""
    import numpy as np
    tp = np.concatenate([np.asarray(t) for t in tp_batches], 0)
    fp = np.concatenate([np.asarray(t) for t in fp_batches], 0)
    fn = np.concatenate([np.asarray(t) for t in fn_batches], 0)
    raw = np.array(names_per_batch).flatten()
    out = {}
    for subject in np.unique(raw):
        idx = np.where(raw == subject)
        i, j, k = tp[idx].sum(0), fp[idx].sum(0), fn[idx].sum(0)
        with np.errstate(invalid="ignore", divide="ignore"):
            iou, dc = i / (i + j + k), 2 * i / (2 * i + j + k)
        out[str(subject)] = {f"mask_{c + 1}": {"IoU": float(iou[c]), "Dice": float(dc[c])} for c in range(len(iou))}
    return out


def lwf_loss_value(base_loss, pred_logits, target_logits, temperature=2.0):
    """deep_supervision.py:201-214 -- KL terms are added to the value; they carry no gradient because
    the predictions are detached (lwf/nnUNetTrainerLWF.py:343)."""
    l = base_loss
    for idx, t in enumerate(target_logits):
        l = l + losses.lwf_distillation(pred_logits[idx].detach(), t.detach(), temperature)
    return l


def _with_head(net, head_state):
    """Load one task's head tensors (``seg_outputs.*``) into the running model (MultiHead_Module.assemble_model)."""
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n in head_state:
                p.copy_(head_state[n])


def lwf_target_logits(net, heads, gen, num_batches):
    """utilities/helpful_functions.py:207-266: for each head IN TURN, ``num_batches`` consecutive batches of ``gen`` go
    through body + that head; the full-resolution logits are kept.  ``heads``: OrderedDict task -> {name: tensor}."""
    keep = {n: p.detach().clone() for n, p in net.named_parameters() if n.startswith("seg_outputs.")}
    out = OrderedDict()
    for task, hs in heads.items():
        _with_head(net, hs)
        out[task] = []
        for _ in range(num_batches):
            b = next(gen)
            with torch.no_grad():
                out[task].append(net(torch.as_tensor(b["data"]))[0].detach().clone())
    _with_head(net, keep)
    return out


def lwf_iteration(net, opt, gen, heads, target_logits, batch_idx, weights, temperature=2.0, clip=12.0):
    """lwf/nnUNetTrainerLWF.py:298-370, the LwF branch (not freeze_run, not do_val, more than one head), as the code runs:
    every ``tee(data_generator, 1)[0]`` copy pulls from the generator ITSELF, so head j is evaluated on batch k+j, the
    network is trained on batch k+T, and batch k+T+1 is skipped (:361) -- T+2 batches per iteration.  ``heads``: OrderedDict
    task -> head state, the LAST one being the task in training, whose tensors live in ``net``.  The KL terms add to the
    value only (predictions detached, :343).  Returns the loss value."""
    tasks = list(heads.keys())
    cur = {n: p.detach().clone() for n, p in net.named_parameters() if n.startswith("seg_outputs.")}
    preds, targets = [], []
    for task in tasks:
        _with_head(net, cur if task == tasks[-1] else heads[task])
        b = next(gen)
        with torch.no_grad():
            preds.append(net(torch.as_tensor(b["data"]))[0].detach())
        if task != tasks[-1]:
            targets.append(target_logits[task][batch_idx % 250])
    _with_head(net, cur)
    b = next(gen)
    data, target = torch.as_tensor(b["data"]), [torch.as_tensor(t) for t in b["target"]]
    opt.zero_grad()
    out = net(data)
    l = lwf_loss_value(losses.multiple_output_loss(out, target, weights), preds, targets, temperature)
    l.backward()
    torch.nn.utils.clip_grad_norm_(net.parameters(), clip)
    opt.step()
    next(gen)
    return float(l.detach())


def lwf_iteration_same_batch(net, opt, gen, heads, target_logits, batch_idx, weights, temperature=2.0, clip=12.0):
    """The flag-fixed variant of ``lwf_iteration`` (product: ``nnUNetTrainerLWF.same_batch_predictions``): the paper's
    distillation -- every head's prediction is taken on the TRAINING batch itself, ONE batch per iteration; otherwise the
    reference's arithmetic (value-only KL of deep_supervision.py:194-196 against the stored teacher logits, clip 12, SGD)."""
    tasks = list(heads.keys())
    cur = {n: p.detach().clone() for n, p in net.named_parameters() if n.startswith("seg_outputs.")}
    b = next(gen)
    data, target = torch.as_tensor(b["data"]), [torch.as_tensor(t) for t in b["target"]]
    preds, targets = [], []
    for task in tasks[:-1]:
        _with_head(net, heads[task])
        with torch.no_grad():
            preds.append(net(data)[0].detach())
        targets.append(target_logits[task][batch_idx % 250])
    _with_head(net, cur)
    opt.zero_grad()
    out = net(data)
    preds.append(out[0].detach())
    l = lwf_loss_value(losses.multiple_output_loss(out, target, weights), preds, targets, temperature)
    l.backward()
    torch.nn.utils.clip_grad_norm_(net.parameters(), clip)
    opt.step()
    return float(l.detach())


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
