"""Loss objects of the hot path with the reference's own interfaces, computed by the fused HIP kernels.

  * ``DC_and_CE_loss``        -- upstream nnunet loss constructed at multihead/nnUNetTrainerMultiHead.py:1385
  * ``MultipleOutputLoss2``   -- upstream deep-supervision wrapper (weights recipe MH.py:1373-1383)
  * ``MultipleOutputLossEWC`` -- nnunet_ext/training/loss_functions/deep_supervision.py:15-83
  * ``MultipleOutputLossLWF`` -- nnunet_ext/training/loss_functions/deep_supervision.py:138-214
Loss contract: ``loss(x: tuple[Tensor], y: list[Tensor]) -> 0-dim Tensor`` taking part in
autograd; mutators ``update_ewc_params``, ``update_network_params``, ``update_logits``.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
from torch import nn

from . import native as nat


def ds_loss_weights(net_numpool: int) -> np.ndarray:
    """multihead/nnUNetTrainerMultiHead.py:1377-1383."""
    weights = np.array([1 / (2 ** i) for i in range(net_numpool)])
    mask = np.array([True] + [True if i < net_numpool - 1 else False for i in range(1, net_numpool)])
    weights[~mask] = 0
    return weights / weights.sum()


def _world_size(group=None):
    import torch.distributed as dist
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def _all_reduce_sum(t, group=None):
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


class _DiceCEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, batch_dice, smooth, group=None):
        logits = logits.contiguous()
        N, K = logits.shape[:2]
        V = logits[0, 0].numel()
        labels = target.reshape(N, V).to(logits.device, torch.float32).contiguous()
        ws = torch.empty(nat.query("lnn_dice_ce_ws_doubles", N, K), dtype=torch.float64, device=logits.device)
        out = torch.empty(1, device=logits.device)
        nat.call("lnn_dice_ce_fwd", logits, labels, N, K, V, int(batch_dice), float(smooth), out, ws)
        world = 1
        if batch_dice:
            # with batch Dice the tp/fp/fn sums run over the GLOBAL batch: one small all-reduce inside
            # the loss forward; the same global sums are reused by backward
            world = _world_size(group)           # the trainer's process group, not the default one
            if world > 1:
                _all_reduce_sum(ws[:N * K * 3], group)
                nat.call("lnn_dice_ce_loss_from_totals", ws, N, K, V, 1, float(smooth), out)
        ctx.save_for_backward(logits, labels, ws)
        ctx.cfg = (N, K, V, int(batch_dice), float(smooth), float(world))
        return out[0]

    @staticmethod
    def backward(ctx, g):
        logits, labels, ws = ctx.saved_tensors
        N, K, V, bd, smooth, world = ctx.cfg
        dl = torch.empty_like(logits)
        # the upstream scalar gradient (deep-supervision weight x loss scale) rides in as gscale; the global Dice term's
        # gradient is scaled by the world size because the gradient all-reduce averages over the ranks
        nat.call("lnn_dice_ce_bwd", logits, labels, N, K, V, bd, smooth, ws, 1.0, g.reshape(1).float().contiguous(), world, dl)
        return dl, None, None, None, None


class DC_and_CE_loss(nn.Module):
    """CE + soft Dice, both weight 1 (upstream aggregate="sum"); do_bg must be False as the trainers
    construct it (``{'batch_dice': .., 'smooth': 1e-5, 'do_bg': False}, {}``)."""

    def __init__(self, soft_dice_kwargs, ce_kwargs=None, aggregate="sum"):
        super().__init__()
        assert aggregate == "sum" and not soft_dice_kwargs.get("do_bg", False)
        self.batch_dice = bool(soft_dice_kwargs.get("batch_dice", False))
        self.smooth = float(soft_dice_kwargs.get("smooth", 1e-5))
        self.process_group = None      # set by the trainer: the ranks a batch-Dice exchange runs over

    def forward(self, net_output, target):
        return _DiceCEFunction.apply(net_output, target, self.batch_dice, self.smooth, self.process_group)


class _DSDiceCEFunction(torch.autograd.Function):
    """``MultipleOutputLoss2(DC_and_CE_loss)`` as ONE autograd node: sum_i w_i * (CE_i + Dice_i) over the levels with a
    non-zero weight.  Same kernels as the per-level path; the weighted sum is formed by the levels' finalize launches
    (``lnn_dice_ce_fwd_ds``) and backward hands every level its weight as the host scalar next to the upstream gradient on
    the device -- no per-level multiply / add kernels (4 + 4 + 3 launches per step on a 5-level network)."""

    @staticmethod
    def forward(ctx, batch_dice, smooth, weights, *tensors):
        L = len(weights)
        logits, targets = tensors[:L], tensors[L:]
        dev = logits[0].device
        total = torch.empty(1, device=dev)
        saved, cfgs, first = [], [], True
        for i in range(L):
            if weights[i] == 0:
                continue
            lg = logits[i].contiguous()
            N, K = lg.shape[:2]
            V = lg[0, 0].numel()
            labels = targets[i].reshape(N, V).to(dev, torch.float32).contiguous()
            ws = torch.empty(nat.query("lnn_dice_ce_ws_doubles", N, K), dtype=torch.float64, device=dev)
            out = torch.empty(1, device=dev)
            nat.call("lnn_dice_ce_fwd_ds", lg, labels, N, K, V, int(batch_dice), float(smooth), out, ws, float(weights[i]),
                     total, 0 if first else 1)
            first = False
            saved += [lg, labels, ws]
            cfgs.append((i, N, K, V))
        ctx.save_for_backward(*saved)
        ctx.cfg = (cfgs, int(batch_dice), float(smooth), tuple(float(w) for w in weights), L)
        return total[0]

    @staticmethod
    def backward(ctx, g):
        cfgs, bd, smooth, weights, L = ctx.cfg
        saved = ctx.saved_tensors
        gdev = g.reshape(1).float().contiguous()
        grads = [None] * L
        for j, (i, N, K, V) in enumerate(cfgs):
            lg, labels, ws = saved[3 * j:3 * j + 3]
            dl = torch.empty_like(lg)
            nat.call("lnn_dice_ce_bwd", lg, labels, N, K, V, bd, smooth, ws, weights[i], gdev, 1.0, dl)
            grads[i] = dl
        return (None, None, None) + tuple(grads) + (None,) * L


class MultipleOutputLoss2(nn.Module):
    def __init__(self, loss, weight_factors=None):
        super().__init__()
        self.weight_factors = weight_factors
        self.loss = loss

    def forward(self, x, y):
        assert isinstance(x, (tuple, list)), "x must be either tuple or list"
        assert isinstance(y, (tuple, list)), "y must be either tuple or list"
        weights = [1] * len(x) if self.weight_factors is None else self.weight_factors
        if type(self.loss) is DC_and_CE_loss and weights[0] != 0 and \
                not (self.loss.batch_dice and _world_size(self.loss.process_group) > 1):
            # (with a data-parallel batch Dice every level needs its own exchange + recomputed loss: per-level path below)
            return _DSDiceCEFunction.apply(self.loss.batch_dice, self.loss.smooth, tuple(float(w) for w in weights[:len(x)]),
                                           *x, *y[:len(x)])
        l = weights[0] * self.loss(x[0], y[0])
        for i in range(1, len(x)):
            if weights[i] != 0:          # zero-weight levels are skipped, not multiplied
                l = l + weights[i] * self.loss(x[i], y[i])
        return l


# ------------------------------------------------------------------------------------------------- EWC
class _EWCPenaltyFunction(torch.autograd.Function):
    """base + sum_tasks lambda/2 * sum F_t (theta - theta*_t)^2 over the flat arena.  The node sits BETWEEN the segmentation
    loss and the result -- ``base`` is its input -- so its backward is the FIRST node autograd runs: the penalty gradient
    g * lambda * F_t (theta - theta*_t) is in the flat gradient arena before the network's backward starts, on the same
    stream.  That order is what lets the data-parallel exchange start during backward (parallel.GradAllReducer.progress): every
    bucket it sends already holds the penalty's share -- replica-identical, so sum / world preserves it."""

    @staticmethod
    def forward(ctx, base, anchor, arena, fishers, stars, ewc_lambda, net=None, touched=()):
        # ``anchor``: one penalised parameter that requires grad -- it makes the result differentiable when ``base`` is not (a loss
        # object called on logits that carry no graph); its gradient, like every parameter's, is written into the arena directly
        ctx.net, ctx.touched = net, touched
        ws = torch.empty(nat.query("lnn_flat_reduce_ws_doubles"), dtype=torch.float64, device=arena.theta.device)
        total = base.detach().reshape(()).to(arena.theta.device, torch.float32)
        out = torch.empty(1, device=arena.theta.device)
        for f, s in zip(fishers, stars):
            nat.call("lnn_ewc_penalty_fwd", arena.theta, s, f, arena.size, float(ewc_lambda), out, ws)
            total = total + out[0]
        ctx.arena, ctx.fishers, ctx.stars, ctx.lam = arena, fishers, stars, float(ewc_lambda)
        return total

    @staticmethod
    def backward(ctx, g):
        gdev = g.reshape(1).float().contiguous()
        for f, s in zip(ctx.fishers, ctx.stars):
            nat.call("lnn_ewc_penalty_bwd", ctx.arena.theta, s, f, ctx.arena.size, ctx.lam, 1.0, gdev, ctx.arena.grad)
        if ctx.net is not None:
            # these parameters now HAVE a gradient (torch: .grad is not None), also the zero-weight deep-supervision head
            # the network's own backward never touches: clip_grad_norm_ counts it and SGD steps it (weight decay, momentum,
            # the pull towards theta*) -- optim._ranges must not skip it
            ctx.net.penalty_grad_names = set(getattr(ctx.net, "penalty_grad_names", ())) | set(ctx.touched)
        return g, None, None, None, None, None, None, None


class MultipleOutputLossEWC(MultipleOutputLoss2):
    def __init__(self, loss, weight_factors=None, ewc_lambda=0.4, fisher=dict(), params=dict(), network_params=None,
                 match_sth=False, match=list(), match_true=True):
        super().__init__(loss, weight_factors)
        self.ewc_lambda = ewc_lambda
        self.network_params = network_params
        self.match_case, self.match, self.match_true = match_sth, match, match_true
        self._flat: Dict = {}
        self.update_ewc_params(fisher, params)

    def update_ewc_params(self, fisher, params):
        self.tasks = list(fisher.keys())
        self.fisher = fisher
        self.params = params
        self._flat = {}                                   # flat arenas are rebuilt lazily

    def update_network_params(self, network_params):
        self.network_params = network_params

    def _selected(self, name):
        if not self.match_case:
            return True
        if self.match_true:
            return all(m in name for m in self.match)
        return all(m not in name for m in self.match)

    def _flat_for(self, task, named, arena):
        key = (task, tuple(n for n, _ in named))
        if key not in self._flat:
            F = torch.zeros(arena.size, device=arena.theta.device)
            S = torch.zeros(arena.size, device=arena.theta.device)
            for name, p in named:
                if not self._selected(name):
                    continue
                s = p._lnn_slot
                # a Fisher of shape [1] (param.grad was None, ewc/nnUNetTrainerEWC.py:300-301) broadcasts
                F[s.offset:s.offset + s.numel] = self.fisher[task][name].to(F.device, torch.float32).expand(s.shape).reshape(-1)
                S[s.offset:s.offset + s.numel] = self.params[task][name].to(S.device, torch.float32).reshape(-1)
            self._flat[key] = (F, S)
        return self._flat[key]

    def _regularised(self, base, lam):
        """``base`` + sum over ``self.tasks`` of lam/2 * sum F_t (theta - theta*_t)^2 (``base`` itself when nothing applies)."""
        if len(self.tasks) == 0 or self.network_params is None:
            return base
        fishers, stars, arena, net, touched, anchor = [], [], None, None, set(), None
        for task in self.tasks:
            # deep_supervision.py:65-66: the task loop is outermost and ``network_params`` is whatever the
            # trainer handed over -- a *generator* for the base EWC trainer (ewc/nnUNetTrainerEWC.py:140,247),
            # which is exhausted after the first task (and stays exhausted until update_network_params).
            named = [(n, p) for n, p in self.network_params]
            if not named:
                continue
            net = named[0][1]._lnn_net
            arena = net.arena
            touched |= {n for n, p in named if p.requires_grad and self._selected(n)}
            anchor = next((p for _, p in named if p.requires_grad), named[0][1])
            F, S = self._flat_for(task, named, arena)
            fishers.append(F)
            stars.append(S)
        if not fishers:
            return base
        return _EWCPenaltyFunction.apply(base, anchor, arena, fishers, stars, lam, net, frozenset(touched))

    def forward(self, x, y, reg=True):
        loss = super().forward(x, y)
        if reg:
            loss = self._regularised(loss, self.ewc_lambda)
        return loss


# ------------------------------------------------------------------------------------------------- RW
class MultipleOutputLossRW(MultipleOutputLossEWC):
    """deep_supervision.py:86-135: ``loss + lambda * sum_{task in tasks} sum (F_task + S_task) * (theta - theta*_task)^2``
    with ``tasks = list(fisher)[:-1]`` (the last entry is the task being trained, :106).  Same flat-arena kernels as
    EWC: the importance scores are added to the Fisher arena and lambda is doubled (EWC's kernel carries the 1/2).
    The generator semantics of ``network_params`` are inherited -- and the RW trainer never refreshes it
    (rw/nnUNetTrainerRW.py:122-125,218-229), so in parity mode the penalty is non-zero for exactly one forward."""

    def __init__(self, loss, weight_factors=None, ewc_lambda=0.4, fisher=dict(), params=dict(), parameter_importance=dict(),
                 network_params=None, match_sth=False, match=list(), match_true=True):
        super().__init__(loss, weight_factors, ewc_lambda, fisher, params, network_params, match_sth, match, match_true)
        self.parameter_importance = parameter_importance

    def update_rw_params(self, fisher, params, parameter_importance):
        super().update_ewc_params(fisher, params)
        self.parameter_importance = parameter_importance
        self.tasks = list(self.fisher.keys())[:-1]

    def _flat_for(self, task, named, arena):
        key = (task, tuple(n for n, _ in named))
        if key not in self._flat:
            F, S = super()._flat_for(task, named, arena)
            for name, p in named:
                if not self._selected(name):
                    continue
                s = p._lnn_slot
                F[s.offset:s.offset + s.numel] += self.parameter_importance[task][name].to(F.device, torch.float32).expand(s.shape).reshape(-1)
            self._flat[key] = (F, S)
        return self._flat[key]

    def forward(self, x, y):
        return self._regularised(MultipleOutputLoss2.forward(self, x, y), 2.0 * self.ewc_lambda)


# ------------------------------------------------------------------------------------------------- MiB
class _TargetCEFunction(torch.autograd.Function):
    """scale * mean_v sum_k q_k (lse(x) - x_k): q = one-hot labels (soft=0) or softmax(alpha * teacher) (soft=1)."""

    @staticmethod
    def forward(ctx, x, target, soft, alpha, ignore_index, scale):
        x = x.contiguous()
        N, K = x.shape[:2]
        V = x[0, 0].numel()
        tgt = target.detach().to(x.device, torch.float32).contiguous()
        ws = torch.empty(2, dtype=torch.float64, device=x.device)
        out = torch.empty(1, device=x.device)
        nat.call("lnn_target_ce_fwd", x, tgt, int(soft), N, K, V, float(alpha), int(ignore_index), float(scale), out, ws)
        ctx.save_for_backward(x, tgt, ws)
        ctx.cfg = (int(soft), N, K, V, float(alpha), int(ignore_index), float(scale))
        return out[0]

    @staticmethod
    def backward(ctx, g):
        x, tgt, ws = ctx.saved_tensors
        soft, N, K, V, alpha, ignore, scale = ctx.cfg
        dx = torch.empty_like(x)
        nat.call("lnn_target_ce_bwd", x, tgt, soft, N, K, V, alpha, ignore, scale, ws, 1.0, g.reshape(1).float().contiguous(), dx)
        return dx, None, None, None, None, None


class RobustCrossEntropyLoss(nn.Module):
    """nnunet_ext/training/loss_functions/crossentropy.py:19-25 (upstream CE on ``target[:, 0].long()``) with the
    ``ignore_index`` the MiB / PLOP losses pass (255); mean over the counted voxels."""

    def __init__(self, ignore_index=-100, **_ignored):
        super().__init__()
        self.ignore_index = ignore_index

    def forward(self, input, target):
        return _TargetCEFunction.apply(input, target, 0, 1.0, self.ignore_index, 1.0)


class UnbiasedKnowledgeDistillationLoss(nn.Module):
    """knowledge_distillation.py:3-32 for EQUAL class sets of student and teacher -- the only way the reference uses it
    (every task keeps the label set, deep_supervision.py:393): new_cl = K, so the "background" term is the plain
    background log-probability and the loss is  -1/K * mean sum_k softmax(alpha*t)_k * log_softmax(x)_k."""

    def __init__(self, reduction='mean', alpha=1.):
        super().__init__()
        assert reduction == 'mean', "the reference constructs it with the default reduction (deep_supervision.py:399)"
        self.reduction, self.alpha = reduction, alpha

    def forward(self, inputs, targets, mask=None):
        assert mask is None and inputs.shape[1] == targets.shape[1], "equal class sets, no mask (deep_supervision.py:409-413)"
        return _TargetCEFunction.apply(inputs, targets, 1, self.alpha, -1, 1.0 / inputs.shape[1])


class MultipleOutputLossMiB(MultipleOutputLoss2):
    """deep_supervision.py:383-416: deep-supervised CE (no Dice, ignore_index 255) + for EVERY level i (zero-weight
    ones included, multiplied by 0) ``weights[i] * lkd * UnbiasedKD(x[i], x_o[i])``; ``x_o`` = the previous model's
    outputs, detached by the caller."""

    def __init__(self, alpha=1., lkd=10, weight_factors=None):
        super().__init__(RobustCrossEntropyLoss(ignore_index=255), weight_factors)
        self.lkd, self.alpha = lkd, alpha
        self.lkd_loss = UnbiasedKnowledgeDistillationLoss(alpha=self.alpha)

    def forward(self, x, x_o, y):
        assert isinstance(x_o, (tuple, list)), "x_o must be either tuple or list"
        loss = super().forward(x, y)
        weights = self.weight_factors if self.weight_factors is not None else [1] * len(x)
        for i in range(len(x)):
            loss = loss + weights[i] * self.lkd * self.lkd_loss(x[i], x_o[i])
        return loss


# ------------------------------------------------------------------------------------------------- LwF
def kl_logits(pred, teach, temperature):
    """F.kl_div(log_softmax(pred/T,1), log_softmax(teach/T,1), 'batchmean', log_target=True)
    (deep_supervision.py:194-196) as one fused device reduction; value only."""
    dev = pred.device if pred.is_cuda else torch.device("cuda")
    p = pred.detach().to(dev, torch.float32).contiguous()
    t = teach.detach().to(dev, torch.float32).contiguous()
    N, K = p.shape[:2]
    V = p[0, 0].numel()
    out = torch.empty(1, device=dev)
    ws = torch.empty(nat.query("lnn_kl_logits_ws_doubles", N), dtype=torch.float64, device=dev)
    nat.call("lnn_kl_logits", p, t, N, K, V, float(temperature), out, ws)
    return out[0]


class MultipleOutputLossLWF(MultipleOutputLoss2):
    def __init__(self, loss, weight_factors=None, pred_logits=list(), target_logits=list(), lwf_temperature=2.0):
        super().__init__(loss, weight_factors)
        self.pred_logits, self.target_logits = pred_logits, target_logits
        self.lwf_temperature = lwf_temperature
        self.scale = [item.size(-1) for item in self.target_logits]   # unused by the reference too (DS.py:196)

    def update_logits(self, pred_logits, target_logits):
        self.pred_logits, self.target_logits = pred_logits, target_logits
        self.scale = [item.size(-1) for item in self.target_logits]

    def _distillation_loss(self, y, teacher_scores, scale=None):
        return kl_logits(y, teacher_scores, self.lwf_temperature)

    def forward(self, x, y):
        loss = super().forward(x, y)
        # one KL per OLD head: target_logits has one entry less than pred_logits (the current task)
        for idx, t_logit in enumerate(self.target_logits):
            loss = loss + self._distillation_loss(self.pred_logits[idx], t_logit)
        return loss


# ------------------------------------------------------------------------------------------------- PLOP / POD
def local_POD(h_, h_old, scales, _dist=None, _pod_lambda=0., _num_layers=1):
    """embeddings.py:9-41 on two 5-D device views (any strides; fp16 or fp32): one fused pass, value only.  The
    reference's failure modes are kept: non-square last two dims raise (its ``torch.cat`` of the width- and height-pooled
    halves does), a window of size 0 at the last scale is its AssertionError.  With ``_dist`` (a 1-element device
    tensor) the kernel also folds ``dist = (dist + lambda * pod) / num_layers`` (DS.py:270-276) into the same launch."""
    assert h_.size() == h_old.size(), "The embedding tensors of the current and old model should have the same shape.."
    assert h_.dim() == 5 and h_.dtype == h_old.dtype and h_.stride() == h_old.stride() and h_.is_cuda and h_old.is_cuda
    N, C, D, H, W = h_.shape
    if H != W:
        raise RuntimeError(f"Sizes of tensors must match except in dimension 1 (local POD pools {W // 2}-wide rows with "
                           f"{H // 2}-wide columns; only H == W works, embeddings.py:31-34)")
    for scale in range(scales):
        assert int(W / 2 ** scale) > 0 and int(H / 2 ** scale) > 0, \
            "The number of scales ({}) are too big in such a way that during scale {} either the step size for H ({}) or W ({}) is 0..".format(
                scales, scale, int(H / 2 ** scale), int(W / 2 ** scale))
    ws = torch.empty(2 * N * C * D, device=h_.device)
    out = torch.empty(1, device=h_.device)
    nat.call("lnn_local_pod", h_, h_old, int(h_.dtype == torch.float16), N, C, D, W, *[int(s) for s in h_.stride()], int(scales),
             float(_pod_lambda), int(_num_layers), ws, _dist, out)
    return out[0]


def _dist_loss(old_interm_results, interm_results, pod_lambda, scales, device):
    """DS.py:268-276 / :370-376: ``dist = (dist + lambda * local_POD(layer)) / num_layers`` per layer IN THE LOOP, in the
    order the old model's hooks fired."""
    dist = torch.zeros(1, device=device)
    L = len(old_interm_results)
    for name, h_old in old_interm_results.items():
        local_POD(interm_results[name], h_old, scales, dist, pod_lambda, L)
    return dist[0]


class _NaNIfEmptyCE(RobustCrossEntropyLoss):
    """``F.cross_entropy(..., ignore_index=255, reduction='mean')`` over ZERO counted voxels is 0/0 = NaN in torch; the
    fused kernel reports 0 there (right for MiB, where the case cannot occur).  PLOP reaches it whenever no voxel passes
    the entropy threshold (tests/golden/plop_reference.json:plop_flow_unconfident), and the reference's loss is NaN."""

    def forward(self, input, target):
        out = super().forward(input, target)
        counted = (target != self.ignore_index).any()
        return out + torch.where(counted, torch.zeros_like(out), torch.full_like(out, float("nan")))


class MultipleOutputLossPLOP(nn.Module):
    """deep_supervision.py:217-334.  Per weighted level: pseudo labels of the old model's background-confident voxels
    (one fused kernel: softmax / argmax / entropy / threshold / the two label volumes / num-den counts), then the two
    ignore-index CE terms on the fused CE kernels; plus the value-only local-POD term over every conv output."""

    def __init__(self, nr_classes=1, pod_lambda=1e-2, scales=3, weight_factors=None):
        super().__init__()
        self.scales, self.nr_classes, self.pod_lambda, self.weight_factors = scales, nr_classes, pod_lambda, weight_factors
        self.ce = _NaNIfEmptyCE(ignore_index=255)

    def update_plop_params(self, old_interm_results, interm_results, thresholds, max_entropy):
        self.thresholds, self.max_entropy = thresholds, max_entropy
        self.interm_results, self.old_interm_results = interm_results, old_interm_results
        self.num_layers = len(self.old_interm_results.keys())

    def forward(self, x, x_o, y):
        assert isinstance(x, (tuple, list)), "x must be either tuple or list"
        assert isinstance(x_o, (tuple, list)), "x_o must be either tuple or list"
        assert isinstance(y, (tuple, list)), "y must be either tuple or list"
        weights = [1] * len(x) if self.weight_factors is None else self.weight_factors
        pseudo_loss = weights[0] * self._pseudo_label_loss(x[0], x_o[0], y[0], idx=0)
        for i in range(1, len(x)):
            if weights[i] != 0:
                pseudo_loss = pseudo_loss + weights[i] * self._pseudo_label_loss(x[i], x_o[i], y[i], idx=i)
        dist_loss = _dist_loss(self.old_interm_results, self.interm_results, self.pod_lambda, self.scales, x[0].device)
        del self.thresholds, self.max_entropy, self.interm_results, self.old_interm_results
        return pseudo_loss + dist_loss

    def _pseudo_label_loss(self, x, x_o, y, idx):
        N, K = x.shape[:2]
        assert N > 1, "batch size 1: y.squeeze() drops the batch axis and the reference's CE fails on the shape (DS.py:265)"
        D, H, W = x.shape[2:]
        dev = x.device
        xo = x_o.detach().to(dev, torch.float32).contiguous()
        yy = y.detach().to(dev, torch.float32).contiguous()
        thr = torch.as_tensor(self.thresholds[idx], dtype=torch.float32, device=dev).contiguous()
        lab, pse = torch.empty((N, 1, D, H, W), device=dev), torch.empty((N, 1, D, H, W), device=dev)
        cnt = torch.empty((2, N, W), dtype=torch.int32, device=dev)
        nat.call("lnn_plop_pseudo_labels", xo, yy, thr, float(self.max_entropy), N, K, D, H, W, lab, pse, cnt[0], cnt[1])
        # DS.py:299-304: masks summed over dims (1,2) of (B,D,H,W) -> a (B,W) table of ratios; it multiplies two SCALAR CE
        # terms, so only its mean matters
        factor = (cnt[0].float() / cnt[1].float()).mean()
        return factor * (self.ce(x, pse) + self.ce(x, lab))


class MultipleOutputLossPOD(MultipleOutputLoss2):
    """deep_supervision.py:337-381: the deep-supervised base loss + the value-only local-POD term."""

    def __init__(self, loss, weight_factors=None, pod_lambda=1e-2, scales=3):
        super().__init__(loss, weight_factors)
        self.pod_lambda, self.scales = pod_lambda, scales

    def update_plop_params(self, old_interm_results, interm_results):
        self.old_interm_results, self.interm_results = old_interm_results, interm_results
        self.num_layers = len(self.old_interm_results.keys())

    def forward(self, x, y):
        loss = super().forward(x, y)
        return loss + _dist_loss(self.old_interm_results, self.interm_results, self.pod_lambda, self.scales, x[0].device)
