"""Continual-learning parity (-m gpu): the fused EWC / LwF loss classes against values produced by the REFERENCE's
own classes (tests/golden/*_reference.npz), and the EWC / LwF / Sequential trainer flows against the CPU oracle."""
import json
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import losses as olosses, train as otrain          # noqa: E402
from oracle.unet import OracleGenericUNet                       # noqa: E402
from lifelong_nnunet_amd import get_trainer_class               # noqa: E402
from lifelong_nnunet_amd.engine import ParamArena, ParamSlot    # noqa: E402
from lifelong_nnunet_amd.losses import (DC_and_CE_loss, MultipleOutputLossEWC, MultipleOutputLossLWF,   # noqa: E402
                                        ds_loss_weights)
from lifelong_nnunet_amd.training.network_training.multihead.nnUNetTrainerMultiHead import default_data_provider  # noqa: E402

DEV = "cuda:0"
TOY = {"patch_size": (16, 16, 16), "batch_size": 2, "num_pool": 2, "base_num_features": 8, "num_classes": 3,
       "num_input_channels": 1, "synthetic_period": 4}


def _base():
    return DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False}, {})


def test_ewc_loss_class_matches_reference_values(golden_dir):
    d = np.load(golden_dir + "/ewc_reference.npz")
    names = json.load(open(golden_dir + "/meta.json"))["ewc"]["names"]
    slots = [ParamSlot(n, tuple(d[f"theta_{i}"].shape)) for i, n in enumerate(names)]
    arena = ParamArena(slots, DEV)
    fake_net = types.SimpleNamespace(arena=arena)
    params = []
    for i, s in enumerate(slots):
        p = torch.nn.Parameter(arena.view(s))
        p._lnn_net, p._lnn_slot = fake_net, s
        with torch.no_grad():
            p.copy_(torch.from_numpy(d[f"theta_{i}"]))
        p.grad = arena.view(s, "grad")
        params.append((s.name, p))
    fisher = {t: {n: torch.from_numpy(d[f"fisher_{t}_{i}"]) for i, n in enumerate(names)} for t in ("taskA", "taskB")}
    star = {t: {n: torch.from_numpy(d[f"star_{t}_{i}"]) for i, n in enumerate(names)} for t in ("taskA", "taskB")}
    xs = tuple(torch.from_numpy(d[f"logits_{i}"]).to(DEV) for i in range(2))
    ys = [torch.from_numpy(d[f"target_{i}"]).to(DEV) for i in range(2)]
    lam = float(d["lambda"])
    # generator semantics (what nnUNetTrainerEWC hands over): first task only
    loss = MultipleOutputLossEWC(_base(), d["ds_weights"], lam, fisher, star, iter(params))
    v = loss(xs, ys)
    assert abs(float(v) - float(d["ref_value_generator"])) <= 1e-5 * abs(float(d["ref_value_generator"]))
    arena.grad.zero_()
    v.backward()
    for i, (n, p) in enumerate(params):
        assert torch.allclose(p.grad.cpu(), torch.from_numpy(d[f"grad_generator_{i}"]), rtol=1e-4, atol=1e-6)
    # the generator is now exhausted: no penalty until update_network_params (EWC.py:247)
    v2 = loss(xs, ys)
    assert abs(float(v2) - float(d["base_loss"])) <= 1e-5 * abs(float(d["base_loss"]))
    loss.update_network_params(iter(params))
    assert abs(float(loss(xs, ys)) - float(d["ref_value_generator"])) <= 1e-5 * abs(float(d["ref_value_generator"]))
    # list semantics (ewc_unet variants): every task
    loss.update_network_params(list(params))
    assert abs(float(loss(xs, ys)) - float(d["ref_value_list"])) <= 1e-5 * abs(float(d["ref_value_list"]))
    assert abs(float(loss(xs, ys, reg=False)) - float(d["base_loss"])) <= 1e-5 * abs(float(d["base_loss"]))


def test_lwf_loss_class_matches_reference_values(golden_dir):
    d = np.load(golden_dir + "/lwf_reference.npz")
    xs = tuple(torch.from_numpy(d[f"logits_{i}"]).to(DEV).requires_grad_(True) for i in range(2))
    ys = [torch.from_numpy(d[f"target_{i}"]).to(DEV) for i in range(2)]
    preds = [torch.from_numpy(d[f"pred_{i}"]).to(DEV) for i in range(3)]
    teach = [torch.from_numpy(d[f"teach_{i}"]) for i in range(2)]        # the reference keeps targets on the CPU
    w = ds_loss_weights(2)
    for T in (1, 2):
        loss = MultipleOutputLossLWF(_base(), w, list(), list(), float(T))
        loss.update_logits(preds, teach)
        v = loss(xs, ys)
        assert abs(float(v) - float(d[f"ref_value_T{T}"])) <= 1e-5 * abs(float(d[f"ref_value_T{T}"]))
    # the distillation term carries NO gradient (LWF.py:343): d(loss)/d(logits) == d(base)/d(logits)
    g = torch.autograd.grad(v, xs[0])[0].cpu()
    xo = torch.from_numpy(d["logits_0"]).requires_grad_(True)
    ob = olosses.multiple_output_loss([xo, torch.from_numpy(d["logits_1"])], [t.cpu() for t in ys], w)
    go = torch.autograd.grad(ob, xo)[0]
    assert float((g - go).abs().max()) <= 1e-4 * float(go.abs().max())


def _make_trainer(ext, task, **kw):
    tr = get_trainer_class(ext)("seg_outputs", task, plans=dict(TOY), device=DEV, **kw)
    tr.initialize(True, num_epochs=1)
    tr.num_batches_per_epoch, tr.num_val_batches_per_epoch = 3, 1
    return tr


def _flat(dct, names):
    return torch.cat([dct[n].detach().float().cpu().reshape(-1) for n in names])


def test_ewc_trainer_flow_matches_oracle():
    torch.manual_seed(12345)
    onet = OracleGenericUNet(1, 8, 3, 2)
    tr = _make_trainer("ewc", "taskA")
    tr.network.load_state_dict(onet.state_dict())
    tr.mh_network.update_after_iteration()
    w = olosses.ds_loss_weights(2)
    oopt = otrain.make_optimizer(onet, lr=tr.optimizer.param_groups[0]['lr'])
    # ---- task A on both sides: 3 training iterations, then after_train on the next 3 batches
    tr.run_training("taskA")
    gen = default_data_provider("taskA", "train", TOY)
    olosses_a = []
    for _ in range(3):
        b = next(gen)
        oopt.param_groups[0]['lr'] = 1e-2 * (1 - 0 / 1) ** 0.9
        olosses_a.append(otrain.run_iteration(onet, oopt, b['data'], b['target'], w)[0])
    assert abs(tr.all_tr_losses[0] - np.mean(olosses_a)) <= 1e-4 * abs(np.mean(olosses_a))
    batches = [(b['data'], b['target']) for b in (next(gen) for _ in range(3))]
    ofisher, oparams = otrain.ewc_after_train(onet, oopt, batches, w)
    names = [n for n, _ in onet.named_parameters()]
    assert list(tr.fisher["taskA"].keys()) == names
    # zero-weight deep-supervision head: Fisher = tensor([1]) on both sides (EWC.py:300-301)
    assert tuple(tr.fisher["taskA"]["seg_outputs.0.weight"].shape) == (1,) == tuple(ofisher["seg_outputs.0.weight"].shape)
    real = [n for n in names if n != "seg_outputs.0.weight"]
    fg, fo = _flat(tr.fisher["taskA"], real), _flat(ofisher, real)
    rel_f = float((fg - fo).norm() / fo.norm())
    pg, po = _flat(tr.params["taskA"], names), _flat(oparams, names)
    rel_p = float((pg - po).norm() / po.norm())
    print(f"fisher rel err {rel_f:.3e}  theta* rel err {rel_p:.3e}")
    assert rel_f < 5e-2 and rel_p < 1e-3
    # ---- task B, first iteration: sync the oracle to the GPU state, compare base + EWC penalty
    onet.load_state_dict(tr.network.state_dict())
    of = {"taskA": {n: tr.fisher["taskA"][n].cpu() for n in names}}
    op = {"taskA": {n: tr.params["taskA"][n].cpu() for n in names}}
    tr.reinitialize("taskB")
    tr.mh_network.add_new_task("taskB", use_init=True)
    tr.network = tr.mh_network.assemble_model("taskB")
    onet.load_state_dict(tr.network.state_dict())
    b = next(default_data_provider("taskB", "train", TOY))
    out = onet(b['data'])
    o_total = olosses.multiple_output_loss(out, b['target'], w) + olosses.ewc_penalty(onet.named_parameters(), of, op, 0.4)
    g_total = tr.run_iteration(tr.tr_gen, True)
    print(f"task B iter 0: oracle {float(o_total):.6f} hip {float(g_total):.6f}")
    assert abs(float(g_total) - float(o_total)) <= 1e-4 * abs(float(o_total))
    pen = float(olosses.ewc_penalty(onet.named_parameters(), of, op, 0.4))
    assert pen > 0          # the new head differs from theta* of the old head -> the penalty is live


def test_lwf_trainer_flow():
    tr = _make_trainer("lwf", "taskA")
    tr.run_training("taskA")
    body0 = {n: p.detach().clone() for n, p in tr.mh_network.body.named_parameters()}
    seen = {}
    orig = tr._run_epoch_loop

    def spy():
        if tr.freeze_run and "frozen" not in seen:
            r = orig()
            seen["frozen"] = all(torch.equal(body0[n], p) for n, p in tr.mh_network.body.named_parameters())
            seen["head_moved"] = not torch.equal(tr.mh_network.heads["taskB"].seg_outputs._modules["1"].weight,
                                                 tr.mh_network.heads["taskA"].seg_outputs._modules["1"].weight)
            return r
        return orig()
    tr._run_epoch_loop = spy
    tr.run_training("taskB")
    assert seen["frozen"] and seen["head_moved"]               # phase 1: body frozen, new head trained
    assert set(tr.target_logits.keys()) == {"taskA", "taskB"} and all(len(v) == 3 for v in tr.target_logits.values())
    assert tr.batch_idx == 3 + 1          # 3 training iterations + the epoch's validation iteration (LWF.py:303: only freeze_run / do_val bypass)
    # phase 3 value = base + KL(old head on the current body || stored teacher), recomputed with the oracle
    loss = tr.LwFloss
    assert len(loss.target_logits) == 1 and len(loss.pred_logits) == 2
    kl_o = float(olosses.lwf_distillation(loss.pred_logits[0].cpu(), loss.target_logits[0].cpu(), 2.0))
    from lifelong_nnunet_amd.losses import kl_logits
    assert abs(float(kl_logits(loss.pred_logits[0], loss.target_logits[0], 2.0)) - kl_o) <= 1e-4 * abs(kl_o) + 1e-9
    assert not all(torch.equal(body0[n], p) for n, p in tr.mh_network.body.named_parameters())   # phase 3 trains the body
    # per-head validation while the LwF loss is live (LWF.py on_epoch_end sets do_val around it): plain iterations -- the
    # teacher-logit index must not move, every head is evaluated, do_val is restored
    tr.freeze_run, tr.loss = False, tr.LwFloss
    idx, consumed = tr.batch_idx, []
    orig_provider = tr.data_provider

    def counting_provider(task, split, plans):
        gen = orig_provider(task, split, plans)

        class G:
            def __iter__(self_):
                return self_

            def __next__(self_):
                consumed.append(task)
                return next(gen)
        return G()
    tr.data_provider = counting_provider
    res = tr._perform_validation(num_batches=2)
    assert tr.batch_idx == idx and tr.do_val is False and set(res.keys()) == {"taskA", "taskB"}
    assert len(consumed) == 2 * 2 * 2        # per head and batch: one batch for the names, one for the prediction (tee(), MH.py:812-819)


def test_sequential_trainer_heads_and_checkpoint():
    """What the reference's integration test asserts (test_multi_head_trainer.py:336,366-411), on synthetic tasks."""
    tr = _make_trainer("sequential", "taskA")
    assert [n.split('.')[0] for n, _ in tr.network.named_parameters()][0] == "conv_blocks_context"   # reordered (SEQ.py:65)
    h0 = tr.mh_network.heads["taskA"].seg_outputs._modules["1"].weight.detach().clone()
    tr.run_training("taskA")
    hA = tr.mh_network.heads["taskA"].seg_outputs._modules["1"].weight.detach().clone()
    assert not torch.equal(h0, hA)                                            # head weights changed after training
    ck = tr.save_checkpoint()
    tr.run_training("taskB")
    assert tr.mh_network.active_task == "taskB"
    assert torch.equal(tr.mh_network.heads["taskA"].seg_outputs._modules["1"].weight, hA)   # previous head untouched
    assert not torch.equal(tr.mh_network.heads["taskB"].seg_outputs._modules["1"].weight, hA)  # transferred, then trained
    sd = tr.mh_network.state_dict()
    assert all(torch.equal(sd["model." + n], sd["heads.taskB." + n]) for n in ("seg_outputs.0.weight", "seg_outputs.1.weight"))
    res = tr._perform_validation(num_batches=1)
    assert set(res.keys()) == {"taskA", "taskB"} and all(0.0 <= r["mean_dice"] <= 1.0 for r in res.values())
    tr2 = _make_trainer("sequential", "taskA")
    tr2.load_checkpoint_ram(ck)
    sd2 = tr2.mh_network.state_dict()
    assert all(torch.equal(sd2[k].cpu(), v) for k, v in ck["state_dict"].items())


def test_rw_loss_class_matches_reference_values(golden_dir):
    """MultipleOutputLossRW on the flat arenas vs the reference class executed verbatim (rw_reference.npz)."""
    from lifelong_nnunet_amd.losses import MultipleOutputLossRW
    d = np.load(golden_dir + "/rw_reference.npz")
    meta = json.load(open(golden_dir + "/meta.json"))["rw"]
    names, tasks = meta["names"], meta["tasks"]
    slots = [ParamSlot(n, tuple(d[f"theta_{i}"].shape)) for i, n in enumerate(names)]
    arena = ParamArena(slots, DEV)
    fake_net = types.SimpleNamespace(arena=arena)
    params = []
    for i, s in enumerate(slots):
        p = torch.nn.Parameter(arena.view(s))
        p._lnn_net, p._lnn_slot = fake_net, s
        with torch.no_grad():
            p.copy_(torch.from_numpy(d[f"theta_{i}"]))
        p.grad = arena.view(s, "grad")
        params.append((s.name, p))
    get = lambda key: {t: {n: torch.from_numpy(d[f"{key}_{t}_{i}"]) for i, n in enumerate(names)} for t in tasks}
    fisher, star, imp = get("fisher"), get("star"), get("importance")
    xs = tuple(torch.from_numpy(d[f"logits_{i}"]).to(DEV) for i in range(2))
    ys = [torch.from_numpy(d[f"target_{i}"]).to(DEV) for i in range(2)]
    lam = float(d["lambda"])
    loss = MultipleOutputLossRW(_base(), d["ds_weights"], lam, dict(), dict(), dict(), iter(params))
    loss.update_rw_params(fisher, star, imp)
    assert loss.tasks == tasks[:-1]                       # the task being trained is omitted (DS.py:106)
    v = loss(xs, ys)
    assert abs(float(v) - float(d["ref_value_generator"])) <= 1e-5 * abs(float(d["ref_value_generator"]))
    arena.grad.zero_()
    v.backward()
    for i, (n, p) in enumerate(params):
        assert torch.allclose(p.grad.cpu(), torch.from_numpy(d[f"grad_generator_{i}"]), rtol=1e-4, atol=1e-5)
    # generator exhausted and never refreshed by the RW trainer: base loss from now on
    v2 = loss(xs, ys)
    assert abs(float(v2) - float(d["ref_value_generator_second_call"])) <= 1e-5 * abs(float(v2))
    loss.update_network_params(list(params))
    assert abs(float(loss(xs, ys)) - float(d["ref_value_list"])) <= 1e-5 * abs(float(d["ref_value_list"]))


def test_rw_trainer_flow_matches_oracle():
    """nnUNetTrainerRW (fused lnn_rw_update on the flat arenas) vs oracle.train.rw_update_f_s / rw_finish_task:
    3 training + 1 validation iteration of task A with fisher_update_after=2, then the end-of-task normalisation."""
    torch.manual_seed(12345)
    onet = OracleGenericUNet(1, 8, 3, 2)
    tr = _make_trainer("rw", "taskA", fisher_update_after=2, rw_alpha=0.9, rw_lambda=0.4)
    tr.network.load_state_dict(onet.state_dict())
    tr.mh_network.update_after_iteration()
    w = olosses.ds_loss_weights(2)
    oopt = otrain.make_optimizer(onet, lr=tr.optimizer.param_groups[0]['lr'])
    tr.run_training("taskA")
    # ---- the same on the oracle
    st = otrain.rw_new_task_state(onet)
    gen = default_data_provider("taskA", "train", TOY)
    ol = []
    for _ in range(3):
        b = next(gen)
        ol.append(otrain.run_iteration(onet, oopt, b['data'], b['target'], w)[0])
        otrain.rw_update_f_s(onet, st, alpha=0.9, fisher_update_after=2)
    assert abs(tr.all_tr_losses[0] - np.mean(ol)) <= 1e-4 * abs(np.mean(ol))
    oopt.zero_grad(set_to_none=False)                  # the validation iteration: zero_grad, no backward (MH.py:612)
    otrain.rw_update_f_s(onet, st, alpha=0.9, fisher_update_after=2)
    assert st["count"] == 4 and tr.count == 0          # the trainer resets its counter after the task (:179)
    ofisher, oparams, oscores = otrain.rw_finish_task(onet, st, n_finished=1)
    names = list(ofisher.keys())
    assert list(tr.fisher["taskA"].keys()) == names == list(tr.scores["taskA"].keys())
    assert "seg_outputs.0.weight" in names             # trainable, but without gradient: stays at its initial zero
    fg, fo = _flat(tr.fisher["taskA"], names), _flat(ofisher, names)
    sg, so = _flat(tr.scores["taskA"], names), _flat(oscores, names)
    pg, po = _flat(tr.params["taskA"], list(oparams.keys())), _flat(oparams, list(oparams.keys()))
    rel_f, rel_s, rel_p = [float((a - b).norm() / b.norm()) for a, b in ((fg, fo), (sg, so), (pg, po))]
    print(f"RW fisher rel err {rel_f:.3e}  scores rel err {rel_s:.3e}  theta* rel err {rel_p:.3e}")
    assert rel_f < 5e-2 and rel_s < 5e-2 and rel_p < 1e-3
    assert float(sg.max()) <= 2.0 + 1e-5               # first task: scaled so that the largest score is 2 (:193-197)
    # ---- task B: the penalty is live on the first forward only (generator never refreshed)
    tr.run_training("taskB")
    assert tr.loss.tasks == ["taskA"] and list(tr.fisher.keys()) == ["taskA", "taskB"]
    assert [n for n, _ in tr.loss.network_params] == []


def test_mib_loss_class_matches_reference_values(golden_dir):
    """MultipleOutputLossMiB (fused lnn_target_ce kernels) vs the reference class executed verbatim: value, gradients
    w.r.t. every deep-supervision level, and the distillation term alone at two alphas."""
    from lifelong_nnunet_amd.losses import MultipleOutputLossMiB, RobustCrossEntropyLoss, UnbiasedKnowledgeDistillationLoss
    d = np.load(golden_dir + "/mib_reference.npz")
    xs = tuple(torch.from_numpy(d[f"logits_{i}"]).to(DEV).requires_grad_(True) for i in range(2))
    xo = tuple(torch.from_numpy(d[f"old_logits_{i}"]).to(DEV) for i in range(2))
    ys = [torch.from_numpy(d[f"target_{i}"]).to(DEV) for i in range(2)]
    loss = MultipleOutputLossMiB(alpha=1.0, lkd=10, weight_factors=d["ds_weights"])
    v = loss(xs, xo, ys)
    assert abs(float(v.detach()) - float(d["ref_value"])) <= 1e-5 * abs(float(d["ref_value"]))
    g = torch.autograd.grad(v, xs, allow_unused=True)
    for i in range(2):
        ref = torch.from_numpy(d[f"grad_{i}"])
        gi = torch.zeros_like(ref) if g[i] is None else g[i].cpu()
        assert float((gi - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-9
    for a, key in ((1.0, "ukd_alpha1"), (0.5, "ukd_alpha05")):
        got = float(UnbiasedKnowledgeDistillationLoss(alpha=a)(xs[0].detach(), xo[0]))
        assert abs(got - float(d[key])) <= 1e-5 * abs(float(d[key]))
    # ignore_index: voxels labelled 255 drop out of the mean
    lab = ys[0].clone(); lab[:, :, :2] = 255
    ce = float(RobustCrossEntropyLoss(ignore_index=255)(xs[0].detach(), lab))
    ref = float(torch.nn.functional.cross_entropy(xs[0].detach().cpu(), lab[:, 0].long().cpu(), ignore_index=255))
    assert abs(ce - ref) <= 1e-5 * abs(ref)


def test_mib_trainer_flow_matches_oracle():
    """Task A with the plain loss, task B with CE + unbiased KD against the frozen task-A model: first task-B iteration
    vs the oracle on the same weights / batch."""
    tr = _make_trainer("mib", "taskA", transfer_heads=True)
    tr.run_training("taskA")
    assert tr.network_old is None
    tr.num_batches_per_epoch = 1
    onet_old = OracleGenericUNet(1, 8, 3, 2)
    onet_old.load_state_dict({k: v.cpu() for k, v in tr.network.state_dict().items()})
    b = next(default_data_provider("taskB", "train", TOY))
    # first task-B iteration: the new head is a copy of the old one (transfer_heads) -> new model == old model
    tr.max_num_epochs, tr.epoch = 1, 0
    tr.run_training("taskB")
    assert tr.network_old is not None and not any(p.requires_grad for p in tr.network_old.parameters())
    with torch.no_grad():
        out_old = onet_old(b['data'])
    onet_new = OracleGenericUNet(1, 8, 3, 2)
    onet_new.load_state_dict(onet_old.state_dict())
    # the oracle's new model starts from the same body and (transfer_heads) the same head
    out_new = onet_new(b['data'])
    w = olosses.ds_loss_weights(2)
    o_val = float(olosses.mib_loss(out_new, [o.detach() for o in out_old], b['target'], w, 1.0, 10.0))
    print(f"MiB task B iter 0: oracle {o_val:.6f} hip {tr.all_tr_losses[-1]:.6f}")
    assert abs(tr.all_tr_losses[-1] - o_val) <= 2e-4 * abs(o_val)


def test_ewc_fisher_keeps_loss_scale_flag():
    """The reference's fp16 branch squares the SCALED gradient (EWC.py:287,303: no unscale_ before .pow(2)); the flag
    reproduces it: Fisher = loss_scale^2 x the default (unscaled) Fisher."""
    a = _make_trainer("ewc", "taskA")
    b = _make_trainer("ewc", "taskA", fisher_keeps_loss_scale=True)
    b.network.load_state_dict(a.network.state_dict())
    b.mh_network.update_after_iteration()
    for tr in (a, b):
        tr.tr_gen = tr.data_provider("taskA", "train", tr.plans)        # same batches for both
        tr.fisher["taskA"], tr.params["taskA"] = {}, {}
        tr.after_train()
    scale = a.amp_grad_scaler.get_scale()
    assert scale == b.amp_grad_scaler.get_scale() == 65536.0
    names = [n for n, f in a.fisher["taskA"].items() if f.numel() > 1]
    fa, fb = _flat(a.fisher["taskA"], names), _flat(b.fisher["taskA"], names)
    assert float(fa.sum()) > 0
    assert float((fb / scale ** 2 - fa).norm() / fa.norm()) < 1e-3      # fp32 atomics of the weight-gradient kernels: order only
    assert all(float(b.fisher["taskA"][n]) == 1.0 for n in b.fisher["taskA"] if b.fisher["taskA"][n].numel() == 1)
