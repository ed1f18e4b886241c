"""-m gpu: the fp32-STORAGE mode (trainer ``fp16=False`` = the reference's MH.py:632-641 branch / ``--fp32``).

With fp32 activations and fp64 fixed-order accumulation the HIP path follows the reference's CPU arithmetic to round-off:
  * every fp32 kernel entry vs the plain PyTorch CPU fp32 op: relative L2 <= 2e-6 AND elementwise allclose;
  * the EWC and RW trainer flows vs the values the REFERENCE's own trainers produced (tests/golden/trainer_reference.*):
    every loss of both tasks <= 1e-4 (north_star; measured ~1e-6), Fisher / scores / theta* / updated weights <= 1e-4
    relative L2 -- the quantities the fp16-storage mode only reaches to ~1e-2;
  * two identical steps give bit-identical gradients (no atomics anywhere on this path)."""
import json
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import losses as olosses, train as otrain                      # noqa: E402
from oracle.unet import OracleGenericUNet                                   # noqa: E402
from lifelong_nnunet_amd import get_trainer_class, native as nat          # noqa: E402
from lifelong_nnunet_amd.synthetic import make_patch_batch                 # noqa: E402

DEV = "cuda:0"
TOY = {"patch_size": (16, 16, 16), "batch_size": 2, "num_pool": 2, "base_num_features": 8, "num_classes": 3,
       "num_input_channels": 1, "synthetic_period": 4}


def _rand(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def _cl(x, ld=None, off=0):
    n, c = x.shape[:2]
    ld = ld or c
    buf = torch.zeros((n,) + tuple(x.shape[2:]) + (ld,), device=DEV)
    buf[..., off:off + c] = x.permute(0, 2, 3, 4, 1).to(DEV)
    return buf


def _ncdhw(buf, c, off=0):
    return buf[..., off:off + c].permute(0, 4, 1, 2, 3).contiguous().cpu()


class _V:
    def __init__(self, t, off):
        self.t, self.off = t, off

    def data_ptr(self):
        return self.t.data_ptr() + 4 * self.off


def _close(got, exp, rl2=2e-6):
    rel = float((got.double() - exp.double()).norm() / (exp.double().norm() + 1e-30))
    assert rel <= rl2, rel
    assert torch.allclose(got, exp, rtol=1e-4, atol=1e-5 * float(exp.abs().max()))


@pytest.mark.parametrize("N,C,K,D,H,W,k,st", [(2, 8, 16, 6, 9, 7, (1, 3, 3), (1, 2, 2)), (1, 5, 8, 7, 9, 11, (3, 1, 3), (2, 1, 2)),
                                              (1, 2, 8, 5, 6, 7, (3, 3, 1), (1, 1, 1))])
def test_f32_generic_geometry_kernels(N, C, K, D, H, W, k, st):
    """fp32 parity kernels with per-axis kernel extents / strides (lnn_f32_*_g) vs torch's CPU ops."""
    x = _rand((N, C, D, H, W), 1).requires_grad_(True)
    w = _rand((K, C) + k, 2, 0.2).requires_grad_(True)
    b = _rand((K,), 3)
    y = F.conv3d(x, w, b, stride=st, padding=tuple(a // 2 for a in k))
    dy = _rand(y.shape, 4)
    y.backward(dy)
    xb, yb = _cl(x.detach(), ld=C + 3, off=2), torch.zeros((N,) + tuple(y.shape[2:]) + (K,), device=DEV)
    nat.call("lnn_f32_conv3d_fwd_g", _V(xb, 2), C + 3, w.detach().to(DEV), b.to(DEV), yb, K, N, D, H, W, C, K, *k, *st)
    _close(_ncdhw(yb, K), y.detach())
    dyb, dxb = _cl(dy), torch.zeros((N, D, H, W, C), device=DEV)
    nat.call("lnn_f32_conv3d_dgrad_g", dyb, K, w.detach().to(DEV), dxb, C, N, D, H, W, C, K, *k, *st, 0)
    _close(_ncdhw(dxb, C), x.grad)
    dw = torch.zeros((K, C) + k, device=DEV)
    nat.call("lnn_f32_conv3d_wgrad_g", _V(xb, 2), C + 3, dyb, K, dw, N, D, H, W, C, K, *k, *st)
    _close(dw.cpu(), w.grad)
    # transposed convolution with kernel == stride
    wt = _rand((C, K) + st, 6, 0.2).requires_grad_(True)
    x2 = _rand((N, C, D, H, W), 7).requires_grad_(True)
    yt = F.conv_transpose3d(x2, wt, None, stride=st)
    dyt = _rand(yt.shape, 8)
    yt.backward(dyt)
    x2b, ytb = _cl(x2.detach()), torch.zeros((N,) + tuple(yt.shape[2:]) + (K,), device=DEV)
    nat.call("lnn_f32_convT3d_fwd_g", x2b, C, wt.detach().to(DEV), ytb, K, N, D, H, W, C, K, *st)
    _close(_ncdhw(ytb, K), yt.detach())
    dytb, dx2b = _cl(dyt), torch.zeros((N, D, H, W, C), device=DEV)
    nat.call("lnn_f32_convT3d_dgrad_g", dytb, K, wt.detach().to(DEV), dx2b, C, N, D, H, W, C, K, *st, 0)
    _close(_ncdhw(dx2b, C), x2.grad)
    dwt = torch.zeros((C, K) + st, device=DEV)
    nat.call("lnn_f32_convT3d_wgrad_g", x2b, C, dytb, K, dwt, N, D, H, W, C, K, *st)
    _close(dwt.cpu(), wt.grad)
    # the multi-channel image casts
    img = _rand((N, C, D, H, W), 9)
    dst = torch.full((N, D, H, W, C + 3), 5.0, device=DEV)
    nat.call("lnn_f32_image_to_cl", img.to(DEV), dst, N, C, D * H * W, C + 3)
    assert torch.equal(dst[..., :C].permute(0, 4, 1, 2, 3).cpu(), img) and torch.all(dst[..., C:] == 5.0)
    dsth = torch.zeros((N, D, H, W, 16), dtype=torch.float16, device=DEV)
    nat.call("lnn_image_to_cl_h", img.to(DEV), dsth, N, C, D * H * W, 16)
    assert torch.equal(dsth[..., :C].permute(0, 4, 1, 2, 3).cpu(), img.half()) and torch.all(dsth[..., C:] == 0)


@pytest.mark.parametrize("N,C,K,D,H,W,s", [(2, 8, 16, 6, 9, 7, 1), (1, 16, 8, 7, 9, 11, 2), (1, 1, 8, 5, 6, 7, 1), (2, 24, 24, 4, 4, 6, 2)])
def test_f32_conv_kernels(N, C, K, D, H, W, s):
    x = _rand((N, C, D, H, W), 1).requires_grad_(True)
    w = _rand((K, C, 3, 3, 3), 2, 0.2).requires_grad_(True)
    b = _rand((K,), 3)
    y = F.conv3d(x, w, b, stride=s, padding=1)
    dy = _rand(y.shape, 4)
    y.backward(dy)
    xb = _cl(x.detach(), ld=C + 3, off=2)
    yb = torch.full((N,) + tuple(y.shape[2:]) + (K + 2,), 7.0, device=DEV)
    nat.call("lnn_f32_conv3d_fwd", _V(xb, 2), C + 3, w.detach().to(DEV), b.to(DEV), _V(yb, 1), K + 2, N, D, H, W, C, K, s)
    _close(_ncdhw(yb, K, 1), y.detach())
    assert torch.all(yb[..., 0] == 7.0)
    dyb = _cl(dy)
    base = _rand(x.shape, 5)
    dxb = _cl(base)
    nat.call("lnn_f32_conv3d_dgrad", dyb, K, w.detach().to(DEV), dxb, C, N, D, H, W, C, K, s, 1)
    _close(_ncdhw(dxb, C), base + x.grad)
    dw = torch.ones((K, C, 3, 3, 3), device=DEV)
    nat.call("lnn_f32_conv3d_wgrad", _V(xb, 2), C + 3, dyb, K, dw, N, D, H, W, C, K, s)
    _close(dw.cpu(), 1.0 + w.grad)
    dw2 = torch.ones((K, C, 3, 3, 3), device=DEV)
    nat.call("lnn_f32_conv3d_wgrad", _V(xb, 2), C + 3, dyb, K, dw2, N, D, H, W, C, K, s)
    assert torch.equal(dw, dw2)


def test_f32_convT_norm_seg_kernels():
    N, C, K, D, H, W = 2, 12, 8, 3, 4, 5
    x = _rand((N, C, D, H, W), 1).requires_grad_(True)
    w = _rand((C, K, 2, 2, 2), 2, 0.3).requires_grad_(True)
    y = F.conv_transpose3d(x, w, stride=2)
    dy = _rand(y.shape, 3)
    y.backward(dy)
    xb, dyb = _cl(x.detach()), _cl(dy)
    yb = torch.zeros((N, 2 * D, 2 * H, 2 * W, K), device=DEV)
    nat.call("lnn_f32_convT3d_k2s2_fwd", xb, C, w.detach().to(DEV), yb, K, N, D, H, W, C, K)
    _close(_ncdhw(yb, K), y.detach())
    dxb = torch.zeros_like(xb)
    nat.call("lnn_f32_convT3d_k2s2_dgrad", dyb, K, w.detach().to(DEV), dxb, C, N, D, H, W, C, K, 0)
    _close(_ncdhw(dxb, C), x.grad)
    dw = torch.zeros((C, K, 2, 2, 2), device=DEV)
    nat.call("lnn_f32_convT3d_k2s2_wgrad", xb, C, dyb, K, dw, N, D, H, W, C, K)
    _close(dw.cpu(), w.grad)
    # InstanceNorm(affine) + LeakyReLU(0.01), forward and backward
    yy = _rand((N, K, 2 * D, 2 * H, 2 * W), 7, 2.0).requires_grad_(True)
    ga = (_rand((K,), 8) * 0.5 + 1).requires_grad_(True)
    be = _rand((K,), 9, 0.3).requires_grad_(True)
    z = F.leaky_relu(F.instance_norm(yy, weight=ga, bias=be, eps=1e-5), 0.01)
    dz = _rand(z.shape, 10)
    z.backward(dz)
    V = 8 * D * H * W
    ybuf, zbuf = _cl(yy.detach()), torch.zeros((N, 2 * D, 2 * H, 2 * W, K), device=DEV)
    mean, rstd = torch.zeros(N * K, device=DEV), torch.zeros(N * K, device=DEV)
    nat.call("lnn_f32_instnorm_lrelu_fwd", ybuf, K, zbuf, K, N, V, K, 1e-5, mean, rstd, ga.detach().to(DEV), be.detach().to(DEV), 0.01)
    _close(_ncdhw(zbuf, K), z.detach())
    dga, dbe = torch.zeros(K, device=DEV), torch.zeros(K, device=DEV)
    ws = torch.zeros(2 * N * K, dtype=torch.float64, device=DEV)
    nat.call("lnn_f32_instnorm_lrelu_bwd", ybuf, K, _cl(dz), K, N, V, K, mean, rstd, ga.detach().to(DEV), be.detach().to(DEV), 0.01,
             dga, dbe, ws)
    _close(_ncdhw(ybuf, K), yy.grad, rl2=1e-5)
    _close(dga.cpu(), ga.grad, rl2=1e-5)
    _close(dbe.cpu(), be.grad, rl2=1e-5)
    # 1x1x1 seg head
    zz = _rand((N, K, D, H, W), 11).requires_grad_(True)
    sw = _rand((3, K, 1, 1, 1), 12, 0.3).requires_grad_(True)
    lg = F.conv3d(zz, sw)
    dl = _rand(lg.shape, 13)
    lg.backward(dl)
    zb = _cl(zz.detach())
    out = torch.zeros((N, 3, D, H, W), device=DEV)
    nat.call("lnn_f32_seg1x1_fwd", zb, K, sw.detach().to(DEV), out, N, D * H * W, K, 3)
    _close(out.cpu(), lg.detach())
    gz, gw = torch.zeros_like(zb), torch.zeros((3, K), device=DEV)
    nat.call("lnn_f32_seg1x1_bwd", zb, K, sw.detach().to(DEV), dl.to(DEV), gz, K, gw, N, D * H * W, K, 3, 0)
    _close(_ncdhw(gz, K), zz.grad)
    _close(gw.cpu(), sw.grad.reshape(3, K))


# ------------------------------------------------------------------------------------------------ trainer flows
@pytest.fixture(scope="module")
def ref(golden_dir):
    return json.load(open(golden_dir + "/trainer_reference.json")), np.load(golden_dir + "/trainer_reference.npz")


def _batches(task_seed, n):
    out = []
    for i in range(n):
        d, t = make_patch_batch(2, (16, 16, 16), 2, seed=task_seed + i)
        out.append({"data": d, "target": t, "keys": [f"case_{task_seed + i}_{b}" for b in range(2)]})
    return out


def _trainer(ext, seeds, n, arr, iters, **kw):
    provider = lambda task, split, plans: iter(_batches(seeds[str(task)] + (0 if split == "train" else 500), n))
    tr = get_trainer_class(ext)("seg_outputs", "taskA", plans=dict(TOY), device=DEV, data_provider=provider, fp16=False, **kw)
    tr.initialize(True, num_epochs=1)
    assert tr.network.storage == "fp32" and tr.amp_grad_scaler.get_scale() == 1.0
    tr.num_batches_per_epoch, tr.num_val_batches_per_epoch = iters, 0
    init = {k[6:]: torch.from_numpy(arr[k]) for k in arr.files if k.startswith("init::")}
    tr.network.load_state_dict(init)
    tr.mh_network.update_after_iteration()
    tr.mh_network.state_init = OrderedDict((k, init[k].to(DEV)) for k in tr.mh_network.state_init)
    return tr


def _rel(arr, key, d, names, sub=7):
    flat = torch.cat([d[n].detach().float().cpu().reshape(-1) for n in names]).numpy()
    exp = arr[key + "::sub"]
    got = flat[::sub]
    assert got.shape == exp.shape, key
    return float(np.linalg.norm(got - exp) / (np.linalg.norm(exp) + 1e-30))


def _record(tr):
    losses = []
    orig = tr.run_iteration
    tr.run_iteration = lambda *a, **k: (lambda v: (losses.append(float(v)), v)[1])(orig(*a, **k))
    return losses


def test_fp32_ewc_flow_matches_reference(ref):
    meta, arr = ref
    e = meta["ewc_flow"]
    names = e["names"]
    tr = _trainer("ewc", {"taskA": 1000, "taskB": 2000}, 6, arr, 3)
    losses = _record(tr)
    tr.run_training("taskA")
    assert np.allclose(losses, e["lossesA"], rtol=1e-4), (losses, e["lossesA"])
    rf, rp = _rel(arr, "ewc::fisherA", tr.fisher["taskA"], names), _rel(arr, "ewc::paramsA", tr.params["taskA"], names)
    print(f"fp32 EWC task A vs reference: loss rel {np.abs(np.array(losses) / np.array(e['lossesA']) - 1).max():.2e} "
          f"fisher rel-L2 {rf:.2e} theta* rel-L2 {rp:.2e}")
    assert rf < 1e-4 and rp < 1e-5
    del losses[:]
    tr.run_training("taskB")
    assert np.allclose(losses, e["lossesB"], rtol=1e-4), (losses, e["lossesB"])
    rf = _rel(arr, "ewc::fisherB", tr.fisher["taskB"], names)
    rt = _rel(arr, "ewc::final_theta", dict(tr.network.named_parameters()), names)
    print(f"fp32 EWC task B vs reference: loss rel {np.abs(np.array(losses) / np.array(e['lossesB']) - 1).max():.2e} "
          f"fisher rel-L2 {rf:.2e} final theta rel-L2 {rt:.2e}")
    assert rf < 1e-4 and rt < 1e-4
    # the zero-weight deep-supervision head: the network's backward never gives it a gradient, the EWC penalty does (a zero
    # one here: theta == theta*), so torch's SGD steps it -- weight decay + momentum shrink it by 2.6e-6 over the 3 steps.
    # A trainer that leaves it frozen ends on theta*_A instead.
    i = names.index("seg_outputs.0.weight")
    got = float(dict(tr.network.named_parameters())["seg_outputs.0.weight"].double().norm())
    exp, frozen = arr["ewc::final_theta::stats"][i, 1], arr["ewc::paramsA::stats"][i, 1]
    assert abs(frozen - exp) > 1e-6 * exp and abs(got - exp) < 0.2 * abs(frozen - exp), (got, exp, frozen)


def test_fp32_rw_flow_matches_reference(ref):
    meta, arr = ref
    r = meta["rw_flow"]
    names, gnames = r["names"], r["stat_names"]
    tr = _trainer("rw", {"taskA": 3000, "taskB": 4000}, r["iters"], arr, r["iters"], fisher_update_after=r["fisher_update_after"],
                  rw_alpha=r["alpha"], rw_lambda=0.4)
    losses = _record(tr)
    tr.run_training("taskA")
    assert np.allclose(losses, r["lossesA"], rtol=1e-4)
    rf, rs = _rel(arr, "rw::fisherA", tr.fisher["taskA"], gnames), _rel(arr, "rw::scoresA", tr.scores["taskA"], gnames)
    print(f"fp32 RW task A vs reference: fisher rel-L2 {rf:.2e} scores rel-L2 {rs:.2e}")
    assert rf < 1e-4 and rs < 1e-3        # the scores divide by 0.5 * F * dtheta^2 + 1e-8: round-off is amplified where that is tiny
    del losses[:]
    tr.run_training("taskB")
    assert np.allclose(losses, r["lossesB"], rtol=1e-4), (losses, r["lossesB"])
    assert _rel(arr, "rw::final_theta", dict(tr.network.named_parameters()), names) < 1e-4


def test_fp32_step_is_bit_reproducible():
    outs = []
    for _ in range(2):
        torch.manual_seed(3)
        tr = get_trainer_class("sequential")("seg_outputs", "taskA", plans=dict(TOY), device=DEV, fp16=False)
        tr.initialize(True, num_epochs=1)
        b = _batches(77, 1)[0]
        tr.run_iteration(iter([b]), True)
        outs.append((tr.network.arena.grad.clone(), tr.network.arena.theta.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


class _Counting:
    def __init__(self, items):
        self.items, self.n = items, 0

    def __iter__(self):
        return self

    def __next__(self):
        b = self.items[self.n % len(self.items)]
        self.n += 1
        return b


def test_fp32_lwf_flow_matches_reference(ref):
    """north_star names "LwF loss values within 1e-4": the fp32-storage mode of nnUNetTrainerLWF against the reference's
    calculate_target_logits (HF.py:207-266) and phase-3 run_iteration (LWF.py:298-370, DS.py:185-214) executed verbatim
    (tests/golden/trainer_reference.*:lwf_flow): teacher logits, T + 2 batches per iteration, loss values, final weights."""
    from lifelong_nnunet_amd.training.network_training.lwf.nnUNetTrainerLWF import calculate_target_logits
    meta, arr = ref
    f = meta["lwf_flow"]
    names = meta["ewc_flow"]["names"]
    tr = _trainer("lwf", {"taskA": 5000, "taskB": 7000}, 2, arr, 2, lwf_temperature=f["T"])
    tr.freeze_run, tr.loss = False, tr.loss_orig
    gA = _Counting(_batches(5000, 2))
    lA = [float(tr.run_iteration(gA, True)) for _ in range(2)]
    assert np.allclose(lA, f["lossesA"], rtol=1e-4), (lA, f["lossesA"])
    tr.mh_network.add_new_task("taskB", use_init=True)
    tr.network = tr.mh_network.assemble_model("taskB", freeze_body=False)
    gT = _Counting(_batches(6000, 6))
    tr.target_logits = calculate_target_logits(tr.mh_network, gT, 3, False)
    assert gT.n == f["teacher_batches_consumed"] and list(tr.target_logits.keys()) == f["teacher_tasks"]
    worst = 0.0
    for t in tr.target_logits:
        for i, lg in enumerate(tr.target_logits[t]):
            exp = arr[f"lwf::teacher_{t}_{i}"]
            got = lg.float().cpu().numpy()[:, :, ::2, ::2, ::2]
            worst = max(worst, float(np.abs(got - exp).max() / np.abs(exp).max()))
    assert worst < 1e-4, worst
    tr.network.train()
    tr.loss, tr.task, tr.batch_idx = tr.LwFloss, "taskB", 0
    gB = _Counting(_batches(7000, 12))
    lB = [float(tr.run_iteration(gB, True)) for _ in range(3)]
    print(f"fp32 LwF vs reference: teacher logits {worst:.2e}, phase-3 losses rel "
          f"{np.abs(np.array(lB) / np.array(f['lossesB']) - 1).max():.2e}")
    assert gB.n == f["batches_consumed_B"] == 12 and tr.batch_idx == f["batch_idx"]
    assert np.allclose(lB, f["lossesB"], rtol=1e-4), (lB, f["lossesB"])
    assert _rel(arr, "lwf::final_theta", dict(tr.network.named_parameters()), names) < 1e-4


def test_fp32_lwf_same_batch_predictions_match_the_oracle(ref):
    """``same_batch_predictions=True`` (the fix behind a flag: every head evaluated on the training batch, one batch and one body
    pass per iteration -- the variant bench.py reports next to the reference-semantics number) against the oracle run with the
    SAME fix (oracle.train.lwf_iteration_same_batch): losses 1e-4, weights 1e-4, one batch consumed per iteration."""
    from collections import OrderedDict
    from lifelong_nnunet_amd.training.network_training.lwf.nnUNetTrainerLWF import calculate_target_logits
    meta, arr = ref
    f = meta["lwf_flow"]
    tr = _trainer("lwf", {"taskA": 5000, "taskB": 7000}, 2, arr, 2, lwf_temperature=f["T"])
    onet = OracleGenericUNet(1, 8, 3, 2)
    onet.load_state_dict({k: v.detach().cpu() for k, v in tr.network.state_dict().items()})
    oopt = otrain.make_optimizer(onet)
    w = olosses.ds_loss_weights(2)
    head0 = {n: p.detach().clone() for n, p in onet.named_parameters() if n.startswith("seg_outputs.")}      # use_init head state
    # task A: two plain iterations on both sides
    tr.freeze_run, tr.loss = False, tr.loss_orig
    gA, ogA = _Counting(_batches(5000, 2)), _Counting(_batches(5000, 2))
    for _ in range(2):
        la = float(tr.run_iteration(gA, True))
        b = next(ogA)
        lo, _ = otrain.run_iteration(onet, oopt, b["data"], b["target"], w)
        assert abs(la - lo) <= 1e-4 * abs(lo)
    headA = {n: p.detach().clone() for n, p in onet.named_parameters() if n.startswith("seg_outputs.")}
    # task B starts from the initial head (use_init); teacher logits of both heads with the task-A body
    tr.mh_network.add_new_task("taskB", use_init=True)
    tr.network = tr.mh_network.assemble_model("taskB", freeze_body=False)
    otrain._with_head(onet, head0)
    heads = OrderedDict([("taskA", headA), ("taskB", head0)])
    tr.target_logits = calculate_target_logits(tr.mh_network, _Counting(_batches(6000, 6)), 3, False)
    oteach = otrain.lwf_target_logits(onet, heads, _Counting(_batches(6000, 6)), 3)
    tr.network.train()
    tr.loss, tr.task, tr.batch_idx = tr.LwFloss, "taskB", 0
    tr.same_batch_predictions = True
    gB, ogB = _Counting(_batches(7000, 4)), _Counting(_batches(7000, 4))
    for it in range(3):
        lb = float(tr.run_iteration(gB, True))
        lo = otrain.lwf_iteration_same_batch(onet, oopt, ogB, heads, oteach, it, w, temperature=f["T"])
        print(f"LwF same-batch iter {it}: oracle {lo:.6f} hip {lb:.6f}")
        assert abs(lb - lo) <= 1e-4 * abs(lo)
    assert gB.n == ogB.n == 3
    osd = onet.state_dict()
    num = sum(float(((v.cpu() - osd[k]) ** 2).sum()) for k, v in tr.network.state_dict().items())
    den = sum(float((v ** 2).sum()) for v in osd.values())
    assert (num / den) ** 0.5 < 1e-4


@pytest.mark.parametrize("transfer", [False, True])
def test_fp32_mib_flow_matches_reference(golden_dir, ref, transfer):
    """fp32-storage nnUNetTrainerMiB vs the reference's own MiB trainer (tests/golden/mib_flow_reference.*, MiB.py:60-182,
    DS.py:383-416): loss values at 1e-4, final weights at 1e-4."""
    meta, arr = ref
    mmeta = json.load(open(golden_dir + "/mib_flow_reference.json"))
    marr = np.load(golden_dir + "/mib_flow_reference.npz")
    f = mmeta["mib_flow_" + ("transfer" if transfer else "init")]
    tr = _trainer("mib", f["seeds"], 4, arr, 2, transfer_heads=transfer, mib_alpha=f["alpha"], mib_lkd=f["lkd"])
    losses = _record(tr)
    tr.run_training("taskA")
    assert np.allclose(losses, f["lossesA"], rtol=1e-4), (losses, f["lossesA"])
    del losses[:]
    tr.num_batches_per_epoch = 3
    tr.run_training("taskB")
    print(f"fp32 MiB task B vs reference ({'transfer' if transfer else 'init'} head): rel "
          f"{np.abs(np.array(losses) / np.array(f['lossesB']) - 1).max():.2e}")
    assert np.allclose(losses, f["lossesB"], rtol=1e-4), (losses, f["lossesB"])
    rt = _rel(marr, "mib_" + ("transfer" if transfer else "init") + "::final_theta", dict(tr.network.named_parameters()), f["names"])
    assert rt < 1e-4, rt


def test_fp32_rehearsal_ewc_flow_matches_oracle_including_its_fisher():
    """BASELINE configs[4] in the fp32-storage mode: task A, then task B on the fused case list with the EWC penalty.  The
    oracle replays BOTH tasks on the CPU with its OWN Fisher / theta* (oracle.train.ewc_after_train: the reference's
    after_train restated, pinned by trainer_reference.*:ewc_flow) -- the composite's Fisher is compared, not borrowed --
    loss values at 1e-4, Fisher at 1e-4, final weights at 1e-4."""
    from oracle import losses as olosses, train as otrain
    from oracle.unet import OracleGenericUNet
    from lifelong_nnunet_amd.training.network_training.rehearsal.nnUNetTrainerRehearsal import RehearsalPatchGenerator
    torch.manual_seed(12345)
    tr = get_trainer_class("rehearsal_ewc")("seg_outputs", "taskA", plans=dict(TOY), device=DEV, cases_per_task=16, fp16=False)
    tr.initialize(True, num_epochs=1)
    tr.num_batches_per_epoch, tr.num_val_batches_per_epoch = 3, 0
    sd0 = {k: v.detach().cpu().clone() for k, v in tr.network.state_dict().items()}
    got = _record(tr)
    tr.run_training("taskA")
    lossesA = list(got)
    del got[:]
    # ---- oracle, task A: the same case list / generator seed as the trainer's (RehearsalMixin._generator)
    onet = OracleGenericUNet(1, 8, 3, 2)
    onet.load_state_dict(sd0)
    oopt = otrain.make_optimizer(onet)
    w = olosses.ds_loss_weights(2)
    casesA = list(tr.dataset_tr.keys())
    gen = RehearsalPatchGenerator(casesA, dict(TOY), seed=12345 + tr.fold)
    expA = []
    for _ in range(3):
        b = next(gen)
        expA.append(otrain.run_iteration(onet, oopt, b["data"], b["target"], w)[0])
    after = [next(gen) for _ in range(3)]                        # after_train draws num_batches_per_epoch more batches
    ofisher, oparams = otrain.ewc_after_train(onet, oopt, [(b["data"], b["target"]) for b in after], w)
    assert np.allclose(lossesA, expA, rtol=1e-4), (lossesA, expA)
    names = [n for n, _ in onet.named_parameters()]
    big = [n for n in names if ofisher[n].numel() > 1]
    fa = torch.cat([tr.fisher["taskA"][n].detach().float().cpu().reshape(-1) for n in big])
    fb = torch.cat([ofisher[n].reshape(-1) for n in big])
    rf = float((fa - fb).norm() / fb.norm())
    assert all(tuple(tr.fisher["taskA"][n].shape) == tuple(ofisher[n].shape) for n in names)      # incl. the tensor([1]) entries
    pa = torch.cat([tr.params["taskA"][n].detach().float().cpu().reshape(-1) for n in names])
    pb = torch.cat([oparams[n].reshape(-1) for n in names])
    rp = float((pa - pb).norm() / pb.norm())
    print(f"fp32 rehearsal+EWC: task-A Fisher rel-L2 {rf:.2e}, theta* rel-L2 {rp:.2e}")
    assert rf < 1e-4 and rp < 1e-5
    # ---- task B on the fused list, penalty from the ORACLE's own Fisher / theta*
    snap = {}
    orig_loop = tr._run_epoch_loop

    def loop():
        snap["sd"] = {k: v.detach().cpu().clone() for k, v in tr.network.state_dict().items()}
        return orig_loop()
    tr._run_epoch_loop = loop
    tr.run_training("taskB")
    fused = list(tr.dataset_tr.keys())
    # the oracle continues with ITS weights; only the fresh head is taken from the trainer's snapshot (use_init head)
    osd = onet.state_dict()
    for k in osd:
        if k.startswith("seg_outputs."):
            osd[k] = snap["sd"][k]
    onet.load_state_dict(osd)
    fisher, params = {"taskA": ofisher}, {"taskA": oparams}
    pen = lambda: olosses.ewc_penalty(onet.named_parameters(), fisher, params, 0.4)
    gen = RehearsalPatchGenerator(fused, dict(TOY), seed=12345 + tr.fold)
    expB, mixed = [], 0
    for _ in range(3):
        b = next(gen)
        mixed += any(k.startswith("taskA_") for k in b["keys"])
        expB.append(otrain.run_iteration(onet, oopt, b["data"], b["target"], w, extra_loss=pen)[0])
    print(f"fp32 rehearsal+EWC task B: hip {got} oracle {expB} (mixed batches {mixed}/3)")
    assert mixed >= 1 and np.allclose(got, expB, rtol=1e-4), (got, expB)
    ta = torch.cat([p.detach().float().cpu().reshape(-1) for _, p in tr.network.named_parameters()])
    tb = torch.cat([p.detach().reshape(-1) for _, p in onet.named_parameters()])
    assert float((ta - tb).norm() / tb.norm()) < 1e-4


@pytest.mark.parametrize("fp16", [False, True])
def test_lwf_phase1_matches_reference(golden_dir, fp16):
    """LwF phase 1 (head-only warm-up on a frozen body, LWF.py:189-201) against the REFERENCE's own ``run_iteration`` with
    ``freeze_run = True`` executed on the CPU (oracle/make_goldens_lwf_phase1.py): upstream's plain iteration WITHOUT the
    clip at 12 -- the fixture's gradient norms are 50+, so a clipping trainer fails this test -- then the head refresh.
    fp32 storage: losses / final head at 1e-4; fp16 storage: 2e-3."""
    meta = json.load(open(golden_dir + "/lwf_phase1_reference.json"))
    arr = np.load(golden_dir + "/lwf_phase1_reference.npz")
    assert min(meta["grad_norms_phase1"]) > 24.0
    seeds = meta["seeds"]
    provider = lambda task, split, plans: iter(_batches(seeds[str(task)] + (0 if split == "train" else 500), 3))
    tr = get_trainer_class("lwf")("seg_outputs", "taskA", plans=dict(TOY), device=DEV, data_provider=provider, fp16=fp16, lwf_temperature=2.0)
    tr.initialize(True, num_epochs=1)
    tr.mh_network.add_new_task("taskB", use_init=True)
    # the state the reference's phase 1 started from: running model, both heads
    pre = {k[5:]: torch.from_numpy(arr[k]) for k in arr.files if k.startswith("pre::")}
    tr.network.load_state_dict(pre)
    for t in ("taskA", "taskB"):
        tr.mh_network.heads[t].load_state_dict({k.split("::", 2)[2]: torch.from_numpy(arr[k]) for k in arr.files if k.startswith(f"prehead::{t}::")})
    tr.freeze_run = True
    tr.network = tr.mh_network.assemble_model("taskB", freeze_body=True)
    tr.network.load_state_dict(pre)                       # (assemble_model copied head B in; the fixture's model already carries it)
    frozen = [n for n, p in tr.network.named_parameters() if not p.requires_grad]
    assert frozen == meta["frozen"]
    tr.loss, tr.task = tr.loss_orig, "taskB"
    tr.network.train()
    body_before = {n: p.detach().clone() for n, p in tr.network.named_parameters() if n in frozen}
    gen = _Counting(_batches(seeds["taskB"], 3))
    got, norms = [], []
    for _ in range(3):
        got.append(float(tr.run_iteration(gen, True)))
        norms.append(tr.last_grad_norm)
    tol = 1e-4 if not fp16 else 2e-3
    print(f"LwF phase 1 ({'fp16' if fp16 else 'fp32'} storage): losses {got} ref {meta['losses_phase1']}; grad norms {norms} ref {meta['grad_norms_phase1']}")
    assert gen.n == meta["batches_consumed_phase1"] and tr.batch_idx == meta["batch_idx"]
    assert np.allclose(got, meta["losses_phase1"], rtol=tol), (got, meta["losses_phase1"])
    assert np.allclose(norms, meta["grad_norms_phase1"], rtol=10 * tol)
    after = dict(tr.network.named_parameters())
    assert all(torch.equal(body_before[n], after[n].detach()) for n in frozen)              # the body did not move
    assert _rel(arr, "phase1::final_theta", after, meta["names"]) < tol
    head_names = [n for n, _ in tr.mh_network.heads["taskB"].named_parameters()]
    assert _rel(arr, "phase1::headB", dict(tr.mh_network.heads["taskB"].named_parameters()), head_names) < 10 * tol   # refreshed (LWF.py:308)


@pytest.mark.parametrize("ext", ["ewc", "rw"])
def test_restored_trainer_continues_bit_for_bit(ref, tmp_path, ext):
    """VERDICT r5 task 6 / SURVEY.md 8 row f4: task A -> side data + checkpoint on disk -> a NEW trainer built from the stored record
    (``already_trained_on`` of the checkpoint's ``.pkl``, EWC.py:66-78) + ``load_checkpoint_ram`` -> task B.  Its task-B losses equal
    the losses of the trainer that was never torn down BIT FOR BIT (fp32 storage mode: fixed-order reductions), i.e. the regulariser
    survives the round trip through ``ewc_data/*.pkl`` / ``rw_data/*.pkl`` exactly."""
    import pickle
    meta, arr = ref
    out = tmp_path / "results" / "TaskA_TaskB" / "fold_0"
    seeds = {"taskA": 1000, "taskB": 2000}
    kw = dict(fisher_update_after=2, rw_alpha=0.9, rw_lambda=0.4) if ext == "rw" else {}
    tr = _trainer(ext, seeds, 6, arr, 3, output_folder=str(out), **kw)
    tr.run_training("taskA")
    sub = "ewc_data" if ext == "ewc" else "rw_data"
    assert os.path.isfile(tr.already_trained_on["0"]["fisher_at"]) and sub in tr.already_trained_on["0"]["fisher_at"]
    fname = str(out / "model_final_checkpoint.model")
    tr.save_checkpoint(fname)
    losses = _record(tr)
    tr.run_training("taskB")
    assert len(losses) == 3 and tr.loss.tasks
    # ---- a fresh process would do exactly this
    info = pickle.load(open(fname + ".pkl", "rb"))
    rec = info["init"][12]
    assert rec["0"]["fisher_at"] and rec["0"]["finished_training_on"] == ["taskA"]
    tr2 = _trainer(ext, seeds, 6, arr, 3, output_folder=str(out), already_trained_on=rec, **kw)
    assert list(tr2.fisher.keys()) == ["taskA", "taskB"]      # the first trainer rewrote the files when it finished task B
    for d in (tr2.fisher, tr2.params) + ((tr2.scores,) if ext == "rw" else ()):
        d.pop("taskB", None)                                  # (the files were rewritten when the first trainer finished task B)
    tr2.load_checkpoint_ram(torch.load(fname, weights_only=False))
    tr2.epoch = 0
    losses2 = _record(tr2)
    tr2.run_training("taskB")
    print(f"{ext}: task-B losses un-restored {losses} restored {losses2}")
    assert losses2 == losses


def test_lwf_restore_after_the_freeze_run_recomputes_the_same_teacher_logits(ref, tmp_path):
    """LWF.py:220-239,427-448: ``model_freezed.model`` written at the end of the freeze run + the record in ``already_trained_on``; a new
    trainer built from that record skips the freeze run, loads the saved MultiHead_Module state for the teacher pass only and ends
    with the teacher logits the uninterrupted trainer computed -- bit for bit (fp32 storage), its own weights untouched."""
    import copy
    meta, arr = ref
    out = tmp_path / "results" / "TaskA_TaskB" / "fold_0"
    seeds = {"taskA": 1000, "taskB": 2000}
    tr = _trainer("lwf", seeds, 40, arr, 2, output_folder=str(out))
    tr.run_training("taskA")
    snap = {}
    orig = tr._save_freezed_model

    def wrapped():
        orig()
        snap["rec"] = copy.deepcopy(tr.already_trained_on)
        snap["state"] = {k: v.detach().clone() for k, v in tr.mh_network.state_dict().items()}
    tr._save_freezed_model = wrapped
    from lifelong_nnunet_amd.training.network_training.lwf import nnUNetTrainerLWF as lwf_mod
    seen = {}
    orig_calc = lwf_mod.calculate_target_logits

    def calc(mh, gen, n, fp16=True, gpu_id=0):
        out_ = orig_calc(mh, gen, n, fp16, gpu_id)
        seen.setdefault("logits", {k: [t.clone() for t in v] for k, v in out_.items()})
        return out_
    lwf_mod.calculate_target_logits = calc
    try:
        tr.run_training("taskB")
    finally:
        lwf_mod.calculate_target_logits = orig_calc
    rec = snap["rec"]
    assert rec["0"]["freeze_run_finished"] is True and os.path.isfile(rec["0"]["freezed_model_at"])
    assert rec["0"]["ftasks_at_time_of_checkpoint"] == ["taskA", "taskB"] and rec["0"]["factive_task_at_time_of_checkpoint"] == "taskB"
    assert tr.already_trained_on["0"]["freeze_run_finished"] is False and tr.already_trained_on["0"]["freezed_model_at"] is None
    disk = torch.load(rec["0"]["freezed_model_at"], weights_only=False)
    assert disk["optimizer_state_dict"] is None and all(torch.equal(disk["state_dict"][k], v.cpu()) for k, v in snap["state"].items())
    # ---- restore: same record, DIFFERENT weights in the live network (what a later regular checkpoint would hold)
    tr2 = _trainer("lwf", seeds, 40, arr, 2, output_folder=str(out), already_trained_on=rec)
    assert tr2.freeze_run is False
    tr2.mh_network.add_new_task("taskB", use_init=False)
    with torch.no_grad():
        tr2.network.arena.theta.mul_(0.5)
    tr2.network.mark_params_changed()
    tr2.mh_network.update_after_iteration()
    mine = {k: v.detach().clone() for k, v in tr2.mh_network.state_dict().items()}
    # the uninterrupted trainer's generator stood behind the two freeze-run iterations when phase 2 started
    tr2.task = "taskB"
    tr2.tr_gen = iter(_batches(seeds["taskB"], 40)[2:])
    tr2._load_model_and_update_target_logits()
    assert list(tr2.target_logits.keys()) == list(seen["logits"].keys()) == ["taskA", "taskB"]
    for k in seen["logits"]:
        assert len(tr2.target_logits[k]) == len(seen["logits"][k]) == 2
        for a, b in zip(tr2.target_logits[k], seen["logits"][k]):
            assert torch.equal(a, b)
    for k, v in tr2.mh_network.state_dict().items():
        assert torch.equal(v, mine[k]), k
