"""CPU tests: the oracle against the committed golden fixtures.

``ewc_reference.npz`` / ``lwf_reference.npz`` hold values produced by the REFERENCE's own
MultipleOutputLossEWC / MultipleOutputLossLWF classes (executed verbatim through oracle/make_goldens.py's
shim in the build container); the rest pins the oracle against independent restatements."""
import json
from collections import OrderedDict

import numpy as np
import torch

from oracle import losses, train
from oracle.unet import OracleGenericUNet


def _load(golden_dir, name):
    return np.load(f"{golden_dir}/{name}")


def test_ewc_penalty_matches_reference_generator_and_list(golden_dir):
    d = _load(golden_dir, "ewc_reference.npz")
    names = json.load(open(f"{golden_dir}/meta.json"))["ewc"]["names"]
    theta = [(n, torch.nn.Parameter(torch.from_numpy(d[f"theta_{i}"]))) for i, n in enumerate(names)]
    fisher = {t: {n: torch.from_numpy(d[f"fisher_{t}_{i}"]) for i, n in enumerate(names)} for t in ("taskA", "taskB")}
    star = {t: {n: torch.from_numpy(d[f"star_{t}_{i}"]) for i, n in enumerate(names)} for t in ("taskA", "taskB")}
    xs = [torch.from_numpy(d[f"logits_{i}"]) for i in range(2)]
    ys = [torch.from_numpy(d[f"target_{i}"]) for i in range(2)]
    base = losses.multiple_output_loss(xs, ys, d["ds_weights"])
    assert abs(float(base) - float(d["base_loss"])) <= 1e-6 * abs(float(d["base_loss"]))
    v_gen = base + losses.ewc_penalty(theta, fisher, star, float(d["lambda"]), first_task_only=True)
    v_list = base + losses.ewc_penalty(theta, fisher, star, float(d["lambda"]), first_task_only=False)
    assert abs(float(v_gen) - float(d["ref_value_generator"])) <= 1e-6 * abs(float(d["ref_value_generator"]))
    assert abs(float(v_list) - float(d["ref_value_list"])) <= 1e-6 * abs(float(d["ref_value_list"]))
    assert float(d["ref_value_list"]) > float(d["ref_value_generator"])        # generator: first task only
    v_gen.backward()
    for i, (n, p) in enumerate(theta):
        assert torch.allclose(p.grad, torch.from_numpy(d[f"grad_generator_{i}"]), rtol=1e-5, atol=1e-7)


def test_lwf_value_matches_reference(golden_dir):
    d = _load(golden_dir, "lwf_reference.npz")
    xs = [torch.from_numpy(d[f"logits_{i}"]) for i in range(2)]
    ys = [torch.from_numpy(d[f"target_{i}"]) for i in range(2)]
    preds = [torch.from_numpy(d[f"pred_{i}"]) for i in range(3)]
    teach = [torch.from_numpy(d[f"teach_{i}"]) for i in range(2)]
    w = losses.ds_loss_weights(2)
    base = losses.multiple_output_loss(xs, ys, w)
    for T in (1, 2):
        v = train.lwf_loss_value(base, preds, teach, float(T))
        assert abs(float(v) - float(d[f"ref_value_T{T}"])) <= 1e-6 * abs(float(d[f"ref_value_T{T}"]))
        # an explicit restatement of batchmean KL with log targets
        for i in range(2):
            lt = torch.log_softmax(teach[i] / T, 1); ly = torch.log_softmax(preds[i] / T, 1)
            kl = (lt.exp() * (lt - ly)).sum() / preds[i].shape[0]
            assert abs(float(kl) - float(d[f"kl{i}_T{T}"])) <= 1e-5 * abs(float(d[f"kl{i}_T{T}"]))


def test_dice_ce_against_numpy_restatement(golden_dir):
    d = _load(golden_dir, "dice_ce.npz")
    lg, tg = d["logits"].astype(np.float64), d["target"][:, 0].astype(np.int64)
    N, K = lg.shape[:2]
    e = np.exp(lg - lg.max(1, keepdims=True)); p = e / e.sum(1, keepdims=True)
    onehot = np.stack([(tg == k) for k in range(K)], 1).astype(np.float64)
    ce = -np.mean(np.log(np.take_along_axis(p, tg[:, None], 1)))
    for batch_dice, key in ((False, "loss_sample_dice"), (True, "loss_batch_dice")):
        ax = (0, 2, 3, 4) if batch_dice else (2, 3, 4)
        tp = (p * onehot).sum(ax); fp = (p * (1 - onehot)).sum(ax); fn = ((1 - p) * onehot).sum(ax)
        dc = (2 * tp + 1e-5) / (2 * tp + fp + fn + 1e-5 + 1e-8)
        dc = dc[1:] if batch_dice else dc[:, 1:]
        assert abs((ce - dc.mean()) - float(d[key])) <= 1e-5 * abs(float(d[key]))
    assert (tg == 2).sum() == 0            # the empty-class edge case is in the fixture
    tp, fp, fn = losses.online_dice_counts(torch.from_numpy(d["logits"]), torch.from_numpy(d["target"]))
    assert np.array_equal(tp.numpy(), d["tp"]) and np.array_equal(fp.numpy(), d["fp"]) and np.array_equal(fn.numpy(), d["fn"])
    dice, iou = losses.dice_from_counts(tp, fp, fn)
    assert (dice[:, 1] == 0).all()         # class absent from the labels but predicted: Dice 0 (only 0/0 gives NaN)
    d0, _ = losses.dice_from_counts(torch.zeros(1), torch.zeros(1), torch.zeros(1))
    assert torch.isnan(d0).all()           # 0/0 -> NaN -> subject dropped (MH.py:1015-1022)


def test_ds_weights_and_rehearsal(golden_dir):
    m = json.load(open(f"{golden_dir}/meta.json"))
    assert np.allclose(losses.ds_loss_weights(5), [8 / 15, 4 / 15, 2 / 15, 1 / 15, 0])
    assert np.allclose(losses.ds_loss_weights(3), [2 / 3, 1 / 3, 0])
    assert np.allclose(losses.ds_loss_weights(5), m["ds_weights"]["5"])
    picked = train.rehearsal_sample(m["rehearsal"]["keys"], m["rehearsal"]["perc"], m["rehearsal"]["seed"])
    assert picked == m["rehearsal"]["picked"]
    assert [len(p) for p in picked] == [10, 4]


def test_unet_structure_and_toy_step(golden_dir):
    m = json.load(open(f"{golden_dir}/meta.json"))
    for num_pool, n_params, n_tensors in ((3, 5602944, 62), (5, 31195584, 98)):
        net = OracleGenericUNet(1, 32, 3, num_pool)
        ps = list(net.named_parameters())
        assert sum(p.numel() for _, p in ps) == n_params and len(ps) == n_tensors
    d = _load(golden_dir, "toy_unet_step.npz")
    net = OracleGenericUNet(*m["toy_unet"]["ctor"])
    assert [n for n, _ in net.named_parameters()] == m["toy_unet"]["param_names"]
    net.load_state_dict({k[4:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w0::")})
    opt = train.make_optimizer(net)
    data = torch.from_numpy(d["data"]); tgts = [torch.from_numpy(d[f"target_{i}"]) for i in range(2)]
    lval, outs = train.run_iteration(net, opt, data, tgts, losses.ds_loss_weights(2))
    assert abs(lval - float(d["loss"])) <= 1e-5 * abs(float(d["loss"]))
    assert [tuple(o.shape) for o in outs] == [(2, 3, 16, 24, 16), (2, 3, 8, 12, 8)]      # full resolution first
    for k in d.files:
        if k.startswith("w1::"):
            assert torch.allclose(net.state_dict()[k[4:]], torch.from_numpy(d[k]), rtol=1e-4, atol=1e-6), k


def test_fisher_is_last_batch_only():
    from lifelong_nnunet_amd.synthetic import make_patch_batch
    torch.manual_seed(99)
    net = OracleGenericUNet(1, 8, 3, 2)
    opt = train.make_optimizer(net)
    w = losses.ds_loss_weights(2)
    batches = [make_patch_batch(1, (8, 8, 8), 2, seed=s) for s in (1, 2, 3)]
    fi, pa = train.ewc_after_train(net, opt, batches, w)
    opt.zero_grad()
    losses.multiple_output_loss(net(batches[-1][0]), batches[-1][1], w).backward()
    for n, p in net.named_parameters():
        if p.grad is None:
            assert tuple(fi[n].shape) == (1,) and float(fi[n]) == 1.0          # EWC.py:300-301
        else:
            assert torch.equal(fi[n], p.grad.pow(2))
        assert torch.equal(pa[n], p.data)


def test_rw_penalty_matches_reference_generator_and_list(golden_dir):
    """oracle.losses.rw_penalty == MultipleOutputLossRW executed verbatim (tests/golden/rw_reference.npz)."""
    d = np.load(golden_dir + "/rw_reference.npz")
    meta = json.load(open(golden_dir + "/meta.json"))["rw"]
    names, tasks = meta["names"], meta["tasks"]
    theta = [(n, torch.nn.Parameter(torch.from_numpy(d[f"theta_{i}"]))) for i, n in enumerate(names)]
    get = lambda key: OrderedDict((t, OrderedDict((n, torch.from_numpy(d[f"{key}_{t}_{i}"])) for i, n in enumerate(names))) for t in tasks)
    fisher, star, imp = get("fisher"), get("star"), get("importance")
    xs = [torch.from_numpy(d[f"logits_{i}"]) for i in range(2)]
    ys = [torch.from_numpy(d[f"target_{i}"]) for i in range(2)]
    base = losses.multiple_output_loss(xs, ys, d["ds_weights"])
    lam = float(d["lambda"])
    v_gen = base + losses.rw_penalty(theta, fisher, star, imp, lam, first_task_only=True)
    v_list = base + losses.rw_penalty(theta, fisher, star, imp, lam, first_task_only=False)
    assert abs(float(v_gen) - float(d["ref_value_generator"])) <= 1e-6 * abs(float(d["ref_value_generator"]))
    assert abs(float(v_list) - float(d["ref_value_list"])) <= 1e-6 * abs(float(d["ref_value_list"]))
    # exhausted generator == no parameters: the reference's second call returns the base loss
    v_none = base + losses.rw_penalty([], fisher, star, imp, lam)
    assert abs(float(v_none) - float(d["ref_value_generator_second_call"])) <= 1e-6 * abs(float(v_none))
    # the task being trained (last key) never contributes
    v2 = base + losses.rw_penalty(theta, OrderedDict(list(fisher.items())[:2]), star, imp, lam, first_task_only=False)
    assert abs(float(v2) - float(v_gen)) <= 1e-6 * abs(float(v_gen))
    g = torch.autograd.grad(v_gen, [p for _, p in theta])
    for i in range(len(names)):
        assert torch.allclose(g[i], torch.from_numpy(d[f"grad_generator_{i}"]), rtol=1e-5, atol=1e-7)


def test_rw_running_statistics_restatement():
    """rw_update_f_s / rw_finish_task on a two-parameter toy: hand-computed values of the formulas at
    rw/nnUNetTrainerRW.py:231-265 and :183-204."""
    net = torch.nn.Linear(2, 1, bias=False)
    with torch.no_grad():
        net.weight.copy_(torch.tensor([[1.0, -2.0]]))
    st = train.rw_new_task_state(net)
    net.weight.grad = torch.tensor([[0.5, -1.0]])
    train.rw_update_f_s(net, st, alpha=0.9, fisher_update_after=2)            # count 0: no prev yet
    assert torch.allclose(st["fisher"]["weight"], 0.9 * torch.tensor([[0.25, 1.0]]))
    assert float(st["scores"]["weight"].abs().sum()) == 0 and st["count"] == 1
    with torch.no_grad():
        net.weight.copy_(torch.tensor([[0.8, -1.5]]))
    train.rw_update_f_s(net, st, alpha=0.9, fisher_update_after=2)            # count 1: skipped
    assert st["count"] == 2 and torch.allclose(st["prev_param"]["weight"], torch.tensor([[1.0, -2.0]]))
    net.weight.grad = torch.tensor([[1.0, 1.0]])
    F0 = st["fisher"]["weight"].clone()
    train.rw_update_f_s(net, st, alpha=0.9, fisher_update_after=2)            # count 2: score + EMA
    d_ = torch.tensor([[1.0 - 0.8, -2.0 + 1.5]])
    sc = (torch.tensor([[1.0, 1.0]]) * d_) / (0.5 * F0 * d_.pow(2) + train.RW_EPSILON)
    sc[sc < 0] = 0
    assert torch.allclose(st["scores"]["weight"], sc)
    assert torch.allclose(st["fisher"]["weight"], 0.9 * torch.ones(1, 2) + 0.1 * F0)
    fisher, params, scores = train.rw_finish_task(net, st, n_finished=1)
    mx = st["scores"]["weight"].max()
    assert torch.allclose(scores["weight"], 2 * (st["scores"]["weight"] - mx) / (mx - mx + train.RW_EPSILON))
    assert torch.allclose(fisher["weight"], (st["fisher"]["weight"] - mx) / train.RW_EPSILON)
    _, _, scores2 = train.rw_finish_task(net, st, n_finished=2)
    assert torch.equal(scores2["weight"], st["scores"]["weight"])


def test_mib_loss_matches_reference(golden_dir):
    """oracle.losses.mib_loss / unbiased_kd == MultipleOutputLossMiB / UnbiasedKnowledgeDistillationLoss executed
    verbatim (tests/golden/mib_reference.npz), incl. the class-incremental form (more student than teacher classes)."""
    d = np.load(golden_dir + "/mib_reference.npz")
    xs = [torch.from_numpy(d[f"logits_{i}"]).requires_grad_(True) for i in range(2)]
    xo = [torch.from_numpy(d[f"old_logits_{i}"]) for i in range(2)]
    ys = [torch.from_numpy(d[f"target_{i}"]) for i in range(2)]
    v = losses.mib_loss(xs, xo, ys, d["ds_weights"], 1.0, 10.0)
    assert abs(float(v) - float(d["ref_value"])) <= 1e-6 * abs(float(d["ref_value"]))
    g = torch.autograd.grad(v, xs, allow_unused=True)
    for i in range(2):
        gi = g[i] if g[i] is not None else torch.zeros_like(xs[i])
        assert torch.allclose(gi, torch.from_numpy(d[f"grad_{i}"]), rtol=1e-5, atol=1e-8)
    assert abs(float(losses.unbiased_kd(xs[0].detach(), xo[0], 0.5)) - float(d["ukd_alpha05"])) <= 1e-6 * abs(float(d["ukd_alpha05"]))
    xin = torch.from_numpy(d["x_incremental"])
    assert abs(float(losses.unbiased_kd(xin, xo[0], 1.0)) - float(d["ukd_incremental"])) <= 1e-6 * abs(float(d["ukd_incremental"]))


def test_tiled_predictor_restatement_properties():
    """oracle/inference.py (parity unpinned: upstream's tiled predictor is not under /root/reference) -- properties any
    correct restatement must have: (i) a network with position-independent output gives exactly that probability
    everywhere, whatever the tiling / blending / padding; (ii) for a flip-equivariant (pointwise) network mirroring
    changes nothing; (iii) blended probabilities sum to one; (iv) a single tile is returned unblended."""
    from oracle import inference as oinf

    class Const(torch.nn.Module):
        def forward(self, x):
            out = torch.zeros((x.shape[0], 3) + tuple(x.shape[2:]))
            out[:, 1] = 1.0; out[:, 2] = -0.5
            return (out,)

    class Pointwise(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c = torch.nn.Conv3d(1, 3, 1)
        def forward(self, x):
            return (self.c(x),)

    g = torch.Generator().manual_seed(0)
    vol = torch.randn((1, 21, 30, 17), generator=g).numpy()
    p_const = torch.softmax(torch.tensor([0.0, 1.0, -0.5]), 0).numpy()
    for mirror in (False, True):
        seg, prob = oinf.predict_3d_tiled(Const(), vol, (8, 16, 8), 0.5, mirror, (0, 1, 2), True)
        assert prob.shape == (3, 21, 30, 17) and seg.shape == (21, 30, 17)
        assert np.allclose(prob, p_const[:, None, None, None], atol=1e-6) and (seg == 1).all()
    torch.manual_seed(3)
    net = Pointwise()
    _, p0 = oinf.predict_3d_tiled(net, vol, (8, 16, 8), 0.5, False, (), True)
    _, p1 = oinf.predict_3d_tiled(net, vol, (8, 16, 8), 0.5, True, (0, 1, 2), True)
    _, p2 = oinf.predict_3d_tiled(net, vol, (8, 16, 8), 0.25, True, (2,), False)       # other step, no Gaussian
    ref = torch.softmax(net(torch.from_numpy(vol)[None])[0], 1)[0].detach().numpy()
    for p_ in (p0, p1, p2):
        assert np.allclose(p_, ref, atol=1e-5) and np.allclose(p_.sum(0), 1.0, atol=1e-5)
    # volume smaller than the patch in one axis: zero-padded, one tile there, cropped back
    small = vol[:, :5]
    _, ps = oinf.predict_3d_tiled(net, small, (8, 16, 8), 0.5, False, (), True)
    assert ps.shape == (3, 5, 30, 17) and np.allclose(ps, ref[:, :5], atol=1e-5)
