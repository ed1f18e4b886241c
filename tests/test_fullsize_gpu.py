"""Full-size (-m gpu) checks at BASELINE configs[1] shapes (160x192x160, B=2), where the fp32 CPU oracle would need
minutes per tensor: size-independent PROPERTIES of the HIP path through the C-ABI.

  * adjointness:  <conv(x), dy> == <x, dgrad(dy)> == <w, wgrad(x, dy)>   (fwd / dgrad / wgrad of one layer are three
    views of one trilinear form; any indexing error in a tile border, parity class or tap slot breaks the equality)
    for the stride-1, stride-2 and transposed convolutions of the heaviest levels, evaluated at dy = conv(x) so that
    the form is ||y||^2 (no cancellation: the three values agree to ~1e-6, the tolerance is 1e-4);
  * linearity / exact power-of-two scaling of the conv (fp16 storage, fp32 accumulation: scaling by 2 is exact);
  * InstanceNorm output statistics (mean 0, variance 1 per (sample, channel));
  * Dice+CE: shift invariance of the loss in the logits and sum_k dlogits[k] == 0 per voxel;
  * one complete C2 optimisation step: finite loss / gradient norm, parameters change, loss decreases over 3 steps.
Inner products are taken in fp64 on the device with torch (plumbing, not the product)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from lifelong_nnunet_amd import native as nat                   # noqa: E402
from tests.gpu_utils import DEV, pack_conv_dgrad, pack_conv_fwd, pack_convT_dgrad, pack_convT_fwd   # noqa: E402

N, D, H, W = 2, 160, 192, 160


def _randh(shape, seed, scale=1.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(shape, generator=g, device=DEV) * scale).half()


def _dot(a, b):
    return float((a.double() * b.double()).sum())


def _close(a, b, tol):
    assert abs(a - b) <= tol * max(abs(a), abs(b)), (a, b)


@pytest.mark.parametrize("C,K,s,dims", [(64, 32, 1, (D, H, W)), (32, 32, 1, (D, H, W)), (32, 64, 2, (D, H, W)),
                                         (128, 64, 1, (80, 96, 80)), (320, 320, 2, (10, 12, 10))])
def test_conv_trilinear_form_is_consistent(C, K, s, dims):
    d, h, w_ = dims
    do, ho, wo = [(x - 1) // s + 1 for x in dims]
    x = _randh((N, d, h, w_, C), 1, 0.5)
    w = torch.randn((K, C, 3, 3, 3), generator=torch.Generator(device=DEV).manual_seed(3), device=DEV) * 0.05
    w = w.half().float()                      # the kernels see fp16 weights
    zero_b = torch.zeros(K, device=DEV)
    y = torch.empty((N, do, ho, wo, K), dtype=torch.float16, device=DEV)
    nat.call("lnn_conv3d_fwd", x, C, pack_conv_fwd(w), zero_b, y, K, N, d, h, w_, C, K, s)
    dy = y.clone()
    dx = torch.empty_like(x)
    nat.call("lnn_conv3d_dgrad", dy, K, pack_conv_dgrad(w), dx, C, N, d, h, w_, C, K, s, 0)
    panel = torch.zeros(nat.query("lnn_wgrad_panel_elems", 27, K, C), device=DEV)
    nat.call("lnn_conv3d_wgrad", x, C, dy, K, panel, N, d, h, w_, C, K, s)
    dw = torch.zeros((K, C, 3, 3, 3), device=DEV)
    nat.call("lnn_unpack_wgrad", panel, dw, 27, K, C, C * 27, 27, 1, 1.0, 0)
    f = _dot(y, dy)            # = ||y||^2; fp16 rounding of dx / fp32 atomics order: relative error ~ 2^-12 / sqrt(n)
    assert f > 0
    _close(f, _dot(x, dx), 1e-4)
    _close(f, _dot(w, dw), 1e-4)
    # accumulate flag: dx += dgrad
    nat.call("lnn_conv3d_dgrad", dy, K, pack_conv_dgrad(w), dx, C, N, d, h, w_, C, K, s, 1)
    _close(2 * f, _dot(x, dx), 2e-4)


@pytest.mark.parametrize("C,K,dims", [(64, 32, (80, 96, 80)), (320, 320, (5, 6, 5))])
def test_convT_trilinear_form_is_consistent(C, K, dims):
    d, h, w_ = dims
    x = _randh((N, d, h, w_, C), 1, 0.5)
    w = (torch.randn((C, K, 2, 2, 2), generator=torch.Generator(device=DEV).manual_seed(3), device=DEV) * 0.1).half().float()
    y = torch.empty((N, 2 * d, 2 * h, 2 * w_, K), dtype=torch.float16, device=DEV)
    nat.call("lnn_convT3d_k2s2_fwd", x, C, pack_convT_fwd(w), y, K, N, d, h, w_, C, K)
    dy = y.clone()
    dx = torch.empty_like(x)
    nat.call("lnn_convT3d_k2s2_dgrad", dy, K, pack_convT_dgrad(w), dx, C, N, d, h, w_, C, K, 0)
    panel = torch.zeros(nat.query("lnn_wgrad_panel_elems", 8, C, K), device=DEV)
    nat.call("lnn_convT3d_k2s2_wgrad", x, C, dy, K, panel, N, d, h, w_, C, K)
    dw = torch.zeros((C, K, 2, 2, 2), device=DEV)
    nat.call("lnn_unpack_wgrad", panel, dw, 8, C, K, K * 8, 8, 1, 1.0, 0)
    f = _dot(y, dy)
    assert f > 0
    _close(f, _dot(x, dx), 1e-4)
    _close(f, _dot(w, dw), 1e-4)


def test_conv_power_of_two_scaling_is_exact():
    C, K = 32, 32
    x = _randh((N, D, H, W, C), 5, 0.5)
    w = torch.randn((K, C, 3, 3, 3), generator=torch.Generator(device=DEV).manual_seed(6), device=DEV) * 0.05
    wp, b = pack_conv_fwd(w), torch.zeros(K, device=DEV)
    y1, y2 = torch.empty((N, D, H, W, K), dtype=torch.float16, device=DEV), torch.empty((N, D, H, W, K), dtype=torch.float16, device=DEV)
    nat.call("lnn_conv3d_fwd", x, C, wp, b, y1, K, N, D, H, W, C, K, 1)
    nat.call("lnn_conv3d_fwd", x * 2.0, C, wp, b, y2, K, N, D, H, W, C, K, 1)
    # doubling is exact in fp16 (inputs, no overflow) and in the fp32 accumulation, so the rounded outputs double
    # exactly -- except where y1 is an fp16 subnormal (< 6.1e-5: fewer mantissa bits than its doubled twin)
    normal = y1.abs() >= 6.2e-5
    assert torch.equal((y1 * 2.0)[normal], y2[normal])
    assert float(((y1.float() * 2 - y2.float()).abs()).max()) <= 1.2e-7       # subnormal spacing 2^-24, doubled
    # same launch twice: bit-identical (no atomics in the forward path)
    nat.call("lnn_conv3d_fwd", x, C, wp, b, y2, K, N, D, H, W, C, K, 1)
    assert torch.equal(y1, y2)


def test_instnorm_output_statistics():
    C, V = 32, D * H * W
    y = _randh((N, V, C), 7, 3.0) + 1.5
    z = torch.empty_like(y)
    mean, rstd = torch.empty(N * C, device=DEV), torch.empty(N * C, device=DEV)
    ws = torch.zeros(nat.query("lnn_instnorm_ws_doubles", N, C), dtype=torch.float64, device=DEV)
    nat.call("lnn_instnorm_stats", y, N, V, C, 1e-5, mean, rstd, ws)
    nat.call("lnn_instnorm_lrelu_fwd", y, z, C, N, V, C, mean, rstd, torch.ones(C, device=DEV), torch.zeros(C, device=DEV), 1.0)
    zd = z.double()
    assert float(zd.mean(dim=1).abs().max()) < 2e-3
    assert float((zd.var(dim=1, unbiased=False) - 1).abs().max()) < 3e-3
    ref_mean = y.double().mean(dim=1).reshape(-1)
    assert float((mean.double() - ref_mean).abs().max()) < 1e-5 * float(ref_mean.abs().max()) + 1e-6


def test_dice_ce_shift_invariance_and_gradient_sum():
    K, V = 3, D * H * W
    g = torch.Generator(device=DEV).manual_seed(9)
    logits = torch.randn((N, K, V), generator=g, device=DEV) * 2
    labels = torch.randint(0, K, (N, 1, V), generator=g, device=DEV).float()
    ws = torch.zeros(nat.query("lnn_dice_ce_ws_doubles", N, K), dtype=torch.float64, device=DEV)
    out1, out2 = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    nat.call("lnn_dice_ce_fwd", logits, labels, N, K, V, 0, 1e-5, out1, ws)
    dl = torch.empty_like(logits)
    nat.call("lnn_dice_ce_bwd", logits, labels, N, K, V, 0, 1e-5, ws, 1.0, None, 1.0, dl)
    shift = torch.randn((N, 1, V), generator=g, device=DEV)
    ws2 = torch.zeros_like(ws)
    nat.call("lnn_dice_ce_fwd", (logits + shift).contiguous(), labels, N, K, V, 0, 1e-5, out2, ws2)
    assert abs(float(out1) - float(out2)) <= 2e-6 * abs(float(out1)) + 1e-7
    # softmax-based loss: the gradient is orthogonal to the all-ones direction in every voxel
    assert float(dl.sum(dim=1).abs().max()) <= 1e-6 * float(dl.abs().max()) + 1e-12
    # and the loss of one-hot "perfect" logits approaches -1 (Dice) + 0 (CE)
    perfect = (torch.nn.functional.one_hot(labels[:, 0].long(), K).permute(0, 2, 1).float() * 40).contiguous()
    nat.call("lnn_dice_ce_fwd", perfect, labels, N, K, V, 0, 1e-5, out2, ws2)
    assert abs(float(out2) + 1.0) < 1e-4


def test_c2_training_steps_are_finite_and_learn():
    from lifelong_nnunet_amd.losses import DC_and_CE_loss, MultipleOutputLoss2, ds_loss_weights
    from lifelong_nnunet_amd.network import Generic_UNet
    from lifelong_nnunet_amd.optim import FusedSGD, GradScaler
    from lifelong_nnunet_amd.synthetic import make_patch_batch
    net = Generic_UNet(1, 32, 3, 5, patch_size=(D, H, W), batch_size=N, device=DEV)
    loss_fn = MultipleOutputLoss2(DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False}, {}), ds_loss_weights(5))
    opt, scaler = FusedSGD(net, 1e-2, weight_decay=3e-5), GradScaler()
    data, tgts = make_patch_batch(N, (D, H, W), 5, seed=12345)
    data, tgts = data.to(DEV), [t.to(DEV) for t in tgts]
    theta0 = net.arena.theta.clone()
    losses = []
    for _ in range(3):
        opt.zero_grad()
        l = loss_fn(net(data), tgts)
        scaler.scale(l).backward()
        inv = 1.0 / scaler.get_scale()
        opt.grad_norm_pass(inv)
        opt.step(inv_scale=inv, max_norm=12.0)
        norm, found_inf = opt.read_ctrl()
        scaler.update(found_inf)
        lv = float(l.detach())
        assert lv == lv and abs(lv) < 1e6 and norm == norm and norm > 0 and not found_inf
        losses.append(lv)
    assert losses[-1] < losses[0]
    assert float((net.arena.theta - theta0).abs().max()) > 0


@pytest.mark.parametrize("C,dims", [(32, (D, H, W))])
def test_fused_norm_backward_reduce_at_bench_shapes(C, dims):
    """The data-gradient shape of configs[1] whose epilogue takes pass 1 of the preceding block's InstanceNorm backward
    (igemm_conv_v9.hip, EPI = 2), at its real size and with the AUTOMATIC kernel choice: the fused call must actually be fused
    there, reproduce lnn_conv3d_dgrad_ws bit for bit, and its (sample, channel) sums must equal both the separate reduce pass and an
    fp64 evaluation of sum g / sum g xhat from the stored tensors."""
    d, h, w_ = dims
    K, V = C, d * h * w_
    u = _randh((N, d, h, w_, C), 21, 1.5) + 0.25
    dy = _randh((N, d, h, w_, K), 22, 0.5)
    g = torch.Generator(device=DEV).manual_seed(23)
    wt = torch.randn((K, C, 3, 3, 3), generator=g, device=DEV) * 0.05
    gamma = 1 + 0.3 * torch.randn(C, generator=g, device=DEV); gamma[::7] *= -1
    beta = 0.2 * torch.randn(C, generator=g, device=DEV)
    wp = pack_conv_dgrad(wt)
    mean, rstd = torch.empty(N * C, device=DEV), torch.empty(N * C, device=DEV)
    nws = nat.query("lnn_instnorm_ws_doubles", N, C)
    ws0 = torch.zeros(nws, dtype=torch.float64, device=DEV)
    nat.call("lnn_instnorm_stats", u, N, V, C, 1e-5, mean, rstd, ws0)
    out = []
    for fused in (True, False):
        dx = torch.empty((N, d, h, w_, C), dtype=torch.float16, device=DEV)
        ws = torch.zeros(nws, dtype=torch.float64, device=DEV)
        dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        if fused:
            nat.call("lnn_conv3d_dgrad_in_bwd_sums", dy, K, wp, dx, C, N, d, h, w_, C, K, u, mean, rstd, gamma, beta, 0.01, dg, db, 1.0, ws,
                     None, 0)
            assert nat.lib().lnn_debug_last_dgrad_reduce_fused() == 1
        else:
            nat.call("lnn_conv3d_dgrad_ws", dy, K, wp, dx, C, N, d, h, w_, C, K, 1, 0, None, 0)
            nat.call("lnn_instnorm_lrelu_bwd_sums", u, dx, C, N, V, C, mean, rstd, gamma, beta, 0.01, dg, db, 1.0, ws)
        out.append((dx, ws[:N * C * 3].view(N, C, 3)[..., :2].clone(), dg, db))
    (dx1, s1, dg1, db1), (dx0, s0, dg0, db0) = out
    assert torch.equal(dx0, dx1)
    # fp64 reference of the sums, one sample at a time (plumbing: torch on the device)
    ref = torch.empty_like(s0)
    for n in range(N):
        xh = (u[n].reshape(V, C).double() - mean.view(N, C)[n].double()) * rstd.view(N, C)[n].double()
        gg = dx0[n].reshape(V, C).double() * torch.where(gamma.double() * xh + beta.double() > 0, 1.0, 0.01)
        ref[n, :, 0] = gg.sum(0); ref[n, :, 1] = (gg * xh).sum(0)
        del xh, gg
    scale = ref.abs().amax(dim=(0, 1))
    for s in (s0, s1):
        assert float(((s - ref).abs() / scale).max()) < 1e-5, ((s - ref).abs() / scale).amax(dim=(0, 1))
    assert float((dg1 - dg0).abs().max()) <= 1e-5 * float(dg0.abs().max()) and float((db1 - db0).abs().max()) <= 1e-5 * float(db0.abs().max())


def test_c2_backward_is_the_same_with_and_without_the_fused_reduce():
    """One configs[1] backward with the engine's fused data-gradient + normalisation-reduce calls and one with the separate passes
    (the round-3 path): every parameter gradient agrees to summation order (the fused epilogue only changes the order in which the
    per-(sample, channel) sums of pass 1 are added)."""
    from lifelong_nnunet_amd.losses import DC_and_CE_loss, MultipleOutputLoss2, ds_loss_weights
    from lifelong_nnunet_amd.network import Generic_UNet
    from lifelong_nnunet_amd.optim import FusedSGD
    from lifelong_nnunet_amd.synthetic import make_patch_batch
    net = Generic_UNet(1, 32, 3, 5, patch_size=(D, H, W), batch_size=N, device=DEV)
    loss_fn = MultipleOutputLoss2(DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False}, {}), ds_loss_weights(5))
    opt = FusedSGD(net, 1e-2, weight_decay=3e-5)
    data, tgts = make_patch_batch(N, (D, H, W), 5, seed=4321)
    data, tgts = data.to(DEV), [t.to(DEV) for t in tgts]
    grads = []
    for fuse in (True, False):
        eng = net.engine_for(data)
        eng.fuse_in_bwd_reduce = fuse
        opt.zero_grad()
        l = loss_fn(net(data), tgts)
        (l * 1024.0).backward()
        torch.cuda.synchronize()
        grads.append(net.arena.grad.double().clone())
    g1, g0 = grads
    assert float(g0.norm()) > 0
    assert float((g1 - g0).norm() / g0.norm()) < 2e-4, float((g1 - g0).norm() / g0.norm())
    for slot in net.arena.slots:
        a, b = g1[slot.offset:slot.offset + slot.numel], g0[slot.offset:slot.offset + slot.numel]
        assert float((a - b).norm()) <= 2e-3 * float(b.norm()) + 1e-6 * float(g0.norm()), slot.name


@pytest.mark.timeout(600)
@pytest.mark.parametrize("workload", ["c3", "c4", "c5"])
def test_continual_learning_trainers_step_at_baseline_shapes(workload):
    """BASELINE configs[2..4] at their own sizes (c3 / c5: 160x192x160, c4: 160x160x160 -- the shape no other test touches), in
    the state bench.py times them in: nnUNetTrainerEWC / nnUNetTrainerLWF (phase 3, one old head) / nnUNetTrainerRehearsalEWC on
    their SECOND task.  Two iterations each: finite loss and gradient norm, parameters move, no overflow-skipped step; the
    regulariser is live (EWC penalty > 0 and its value reproduced from the flat arenas with torch in fp64; LwF: a KL term between
    the stored teacher logits and the old head's prediction is part of the loss) and the gradient norm of the EWC step contains
    the penalty's share (it differs from the norm of the same step without it)."""
    import gc
    import bench
    dev = torch.device(DEV)
    tr, plans, ext, desc, extra = bench.build_trainer(workload, dev, 0)
    assert tuple(plans["patch_size"]) == ((160, 160, 160) if workload == "c4" else (D, H, W)) and plans["batch_size"] == 2
    arena = tr.network.arena
    theta0 = arena.theta.clone()
    losses = []
    for _ in range(2):
        l = tr.run_iteration(tr.tr_gen, True)
        losses.append(float(l))
        assert losses[-1] == losses[-1] and abs(losses[-1]) < 1e6
        assert tr.last_grad_norm == tr.last_grad_norm and tr.last_grad_norm > 0 and not tr.last_found_inf
    assert float((arena.theta - theta0).abs().max()) > 0
    if ext in ("ewc", "rehearsal_ewc"):
        named = list(tr.network.named_parameters())
        tr.loss.update_network_params(iter(named))
        zero = torch.zeros((), device=dev, requires_grad=True)
        keep = arena.grad.clone()
        arena.grad.zero_()
        pen = tr.loss._regularised(zero, tr.loss.ewc_lambda)
        pen.backward()
        gpen = float(arena.grad.double().norm())
        arena.grad.copy_(keep)
        tr.loss.update_network_params(tr.network.named_parameters())
        task = list(tr.fisher.keys())[0]
        ref = 0.0
        for n, p in named:
            ref += float((tr.fisher[task][n].double() * (p.detach().double() - tr.params[task][n].double()) ** 2).sum())
        ref *= tr.loss.ewc_lambda / 2
        assert float(pen) > 0 and abs(float(pen) - ref) <= 1e-5 * ref, (float(pen), ref)
        assert gpen > 0
        if workload == "c5":
            assert extra["batches_with_a_rehearsed_case"] > 0          # mixed-task batches (REH.py:105-164)
    else:
        assert len(tr.LwFloss.target_logits) == 1 and len(tr.LwFloss.pred_logits) == 2
        assert tuple(tr.LwFloss.target_logits[0].shape) == (2, 3, 160, 160, 160)
        from lifelong_nnunet_amd.losses import kl_logits
        kl = float(kl_logits(tr.LwFloss.pred_logits[0], tr.LwFloss.target_logits[0], tr.LwFloss.lwf_temperature))
        assert kl == kl and kl >= 0
    del tr
    gc.collect()
    torch.cuda.empty_cache()
