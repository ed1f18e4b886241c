"""Kernel-level parity tests (-m gpu): every C-ABI entry of liblnn_hip.so against the plain PyTorch
CPU fp32 op the reference reaches.  Inputs are pre-rounded to fp16
so the only differences are accumulation order (fp32) and the fp16 rounding of the stored output;
tolerances below are stated per test."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from tests.gpu_utils import (DEV, View, from_cl_h, pack, pack_conv_dgrad, pack_conv_fwd, pack_convT_dgrad,  # noqa: E402
                             pack_convT_fwd, q16, rel_err, to_cl_h)
from lifelong_nnunet_amd import native as nat  # noqa: E402


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return q16(torch.randn(shape, generator=g) * scale)


@pytest.fixture(autouse=True)
def _specialised_kernels_only():
    """The shapes of this file are small volumes, which the isotropic entry points hand to the generic flattened-voxel kernels
    by default (csrc/igemm_gen.hip: at most 4096 output voxels).  These tests pin the SPECIALISED kernels (tile / z-streaming /
    LDS-DMA families); the generic ones have their own file, tests/test_gen_gpu.py."""
    assert nat.lib().lnn_debug_set_gen_mode(0) == 0
    yield
    nat.lib().lnn_debug_set_gen_mode(-1)


def test_tr16_lane_mapping():
    """ds_read_b64_tr_b16 with lane-linear addresses: within each 16-lane group the 16x4 block is
    transposed: lane i receives elements {i + 16 j} of the group's 64-half block (j = 0..3)."""
    out = torch.zeros(256, device=DEV)
    nat.call("lnn_debug_tr16_probe", out)
    got = out.cpu().view(64, 4)
    exp = torch.zeros(64, 4)
    for lane in range(64):
        g, i = lane // 16, lane % 16
        for j in range(4):
            exp[lane, j] = g * 64 + i + 16 * j
    print("tr16 probe rows 0..17:\n", got[:18])
    assert torch.equal(got, exp)


CONV_CASES = [
    # N, C, K, D, H, W, stride
    (2, 32, 32, 8, 16, 8, 1),
    (1, 16, 64, 5, 9, 11, 1),
    (1, 8, 8, 4, 8, 8, 1),
    (1, 64, 32, 6, 10, 9, 1),
    (1, 48, 96, 4, 8, 8, 1),
    (2, 32, 64, 8, 16, 8, 2),
    (1, 16, 32, 7, 9, 11, 2),
    (1, 64, 128, 6, 6, 6, 2),
    (1, 32, 64, 2, 12, 20, 2),      # one output plane: every tile of the z-streaming stride-2 kernels starts a new column
    (2, 64, 64, 9, 17, 5, 2),       # odd extents, two channel panels, ragged tiles in every dimension
]


@pytest.mark.parametrize("N,C,K,D,H,W,s", CONV_CASES)
def test_conv3d_fwd(N, C, K, D, H, W, s):
    x = _rand((N, C, D, H, W), 1)
    w = _rand((K, C, 3, 3, 3), 2, 0.1)
    b = torch.randn(K, generator=torch.Generator().manual_seed(3))
    ref = F.conv3d(x, w, b, stride=s, padding=1)
    xb, xv = to_cl_h(x, ld=C + 8, offset=8)
    Do, Ho, Wo = ref.shape[2:]
    yb = torch.full((N, Do, Ho, Wo, K + 16), 7.0, dtype=torch.float16, device=DEV)
    wp = pack_conv_fwd(w.to(DEV))
    nat.call("lnn_conv3d_fwd", View(xb, 8), C + 8, wp, b.to(DEV), View(yb, 16), K + 16, N, D, H, W, C, K, s)
    got = from_cl_h(yb, K, 16)
    assert rel_err(got, ref) < 2e-3          # fp16 output rounding (2^-11) + fp32 accumulation order
    assert torch.all(yb[..., :16] == 7.0)    # channel-offset view untouched outside [16, 16+K)


@pytest.mark.parametrize("N,C,K,D,H,W,s", CONV_CASES)
@pytest.mark.parametrize("acc", [0, 1])
def test_conv3d_dgrad(N, C, K, D, H, W, s, acc):
    x = _rand((N, C, D, H, W), 1).requires_grad_(True)
    w = _rand((K, C, 3, 3, 3), 2, 0.1)
    y = F.conv3d(x, w, None, stride=s, padding=1)
    dy = _rand(y.shape, 4)
    y.backward(dy)
    ref = x.grad
    dyb, _ = to_cl_h(dy)
    base = _rand((N, C, D, H, W), 5)
    dxb, _ = to_cl_h(base, ld=C + 8)
    wp = pack_conv_dgrad(w.to(DEV))
    nat.call("lnn_conv3d_dgrad", dyb, K, wp, dxb, C + 8, N, D, H, W, C, K, s, acc)
    got = from_cl_h(dxb, C)
    exp = ref + base if acc else ref
    assert rel_err(got, exp) < 3e-3


@pytest.mark.parametrize("N,C,K,D,H,W,s", CONV_CASES)
def test_conv3d_wgrad(N, C, K, D, H, W, s):
    x = _rand((N, C, D, H, W), 1)
    w = _rand((K, C, 3, 3, 3), 2, 0.1).requires_grad_(True)
    y = F.conv3d(x, w, None, stride=s, padding=1)
    dy = _rand(y.shape, 4)
    y.backward(dy)
    xb, _ = to_cl_h(x)
    dyb, _ = to_cl_h(dy)
    panel = torch.zeros(nat.query("lnn_wgrad_panel_elems", 27, K, C), device=DEV)
    nat.call("lnn_conv3d_wgrad", xb, C, dyb, K, panel, N, D, H, W, C, K, s)
    dw = torch.full((K, C, 3, 3, 3), 1.0, device=DEV)
    nat.call("lnn_unpack_wgrad", panel, dw, 27, K, C, C * 27, 27, 1, 0.5, 1)
    assert rel_err(dw.cpu(), 1.0 + 0.5 * w.grad) < 1e-3   # fp32 accumulate of fp16 products


@pytest.mark.parametrize("N,K,D,H,W", [(2, 32, 8, 16, 8), (1, 8, 5, 9, 11), (1, 64, 4, 8, 8)])
def test_conv3d_first_layer(N, K, D, H, W):
    x = _rand((N, 1, D, H, W), 1)
    w = _rand((K, 1, 3, 3, 3), 2, 0.3).requires_grad_(True)
    b = torch.randn(K, generator=torch.Generator().manual_seed(3))
    ref = F.conv3d(x, w, b, padding=1)
    dy = _rand(ref.shape, 4)
    ref.backward(dy)
    xh = x[:, 0].to(DEV).half().contiguous()
    yb = torch.zeros((N, D, H, W, K), dtype=torch.float16, device=DEV)
    wp = torch.zeros(nat.query("lnn_packed_weight_elems", 1, K, 27), dtype=torch.float16, device=DEV)
    nat.call("lnn_pack_weights", w.detach().to(DEV), wp, 1, K, 27, 27, 1, 0)
    nat.call("lnn_conv3d_fwd", xh, 1, wp, b.to(DEV), yb, K, N, D, H, W, 1, K, 1)
    assert rel_err(from_cl_h(yb, K), ref.detach()) < 2e-3
    dyb, _ = to_cl_h(dy)
    panel = torch.zeros(nat.query("lnn_wgrad_panel_elems", 1, K, 27), device=DEV)
    nat.call("lnn_conv3d_wgrad", xh, 1, dyb, K, panel, N, D, H, W, 1, K, 1)
    dw = torch.zeros((K, 1, 3, 3, 3), device=DEV)
    nat.call("lnn_unpack_wgrad", panel, dw, 1, K, 27, 27, 1, 0, 1.0, 0)
    assert rel_err(dw.cpu(), w.grad) < 1e-3


CONVT_CASES = [(2, 64, 32, 4, 8, 4), (1, 16, 8, 3, 5, 6), (1, 320, 320, 2, 3, 2), (1, 32, 64, 4, 8, 8), (1, 64, 64, 1, 5, 9), (2, 128, 64, 5, 3, 11)]


@pytest.mark.parametrize("N,C,K,D,H,W", CONVT_CASES)
def test_convT_fwd_dgrad_wgrad(N, C, K, D, H, W):
    x = _rand((N, C, D, H, W), 1).requires_grad_(True)
    w = _rand((C, K, 2, 2, 2), 2, 0.1).requires_grad_(True)
    ref = F.conv_transpose3d(x, w, None, stride=2)
    dy = _rand(ref.shape, 4)
    ref.backward(dy)
    xb, _ = to_cl_h(x.detach())
    wd = w.detach().to(DEV)
    yb = torch.zeros((N, 2 * D, 2 * H, 2 * W, 2 * K), dtype=torch.float16, device=DEV)
    nat.call("lnn_convT3d_k2s2_fwd", xb, C, pack_convT_fwd(wd), yb, 2 * K, N, D, H, W, C, K)
    assert rel_err(from_cl_h(yb, K), ref.detach()) < 2e-3
    assert torch.all(yb[..., K:] == 0)
    dyb, _ = to_cl_h(dy)
    dxb = torch.zeros((N, D, H, W, C), dtype=torch.float16, device=DEV)
    nat.call("lnn_convT3d_k2s2_dgrad", dyb, K, pack_convT_dgrad(wd), dxb, C, N, D, H, W, C, K, 0)
    assert rel_err(from_cl_h(dxb, C), x.grad) < 3e-3
    panel = torch.zeros(nat.query("lnn_wgrad_panel_elems", 8, C, K), device=DEV)
    nat.call("lnn_convT3d_k2s2_wgrad", xb, C, dyb, K, panel, N, D, H, W, C, K)
    dw = torch.zeros((C, K, 2, 2, 2), device=DEV)
    nat.call("lnn_unpack_wgrad", panel, dw, 8, C, K, K * 8, 8, 1, 1.0, 0)
    assert rel_err(dw.cpu(), w.grad) < 1e-3


@pytest.mark.parametrize("N,C,D,H,W", [(2, 32, 8, 16, 8), (1, 8, 5, 9, 11), (2, 320, 3, 4, 3), (1, 64, 16, 16, 16)])
def test_instnorm_lrelu_fwd_bwd(N, C, D, H, W):
    y = _rand((N, C, D, H, W), 1) * 2 + 0.5
    y = q16(y).requires_grad_(True)
    g = torch.Generator().manual_seed(9)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    z = F.leaky_relu(F.instance_norm(y, weight=gamma, bias=beta, eps=1e-5), 0.01)
    dz = _rand(z.shape, 4)
    z.backward(dz)
    V = D * H * W
    yb, _ = to_cl_h(y.detach())
    mean = torch.empty(N * C, device=DEV); rstd = torch.empty(N * C, device=DEV)
    ws = torch.zeros(nat.query("lnn_instnorm_ws_doubles", N, C), dtype=torch.float64, device=DEV)
    nat.call("lnn_instnorm_stats", yb, N, V, C, 1e-5, mean, rstd, ws)
    yf = y.detach()
    assert rel_err(mean.cpu().view(N, C), yf.mean((2, 3, 4))) < 1e-5
    assert rel_err(rstd.cpu().view(N, C), 1 / torch.sqrt(yf.var((2, 3, 4), unbiased=False) + 1e-5)) < 1e-5
    zb = torch.zeros((N, D, H, W, C + 8), dtype=torch.float16, device=DEV)
    nat.call("lnn_instnorm_lrelu_fwd", yb, zb, C + 8, N, V, C, mean, rstd, gamma.detach().to(DEV), beta.detach().to(DEV), 0.01)
    assert rel_err(from_cl_h(zb, C), z.detach()) < 2e-3
    dzb, _ = to_cl_h(dz, ld=C + 8)
    dg = torch.zeros(C, device=DEV); db = torch.zeros(C, device=DEV); dbias = torch.zeros(C, device=DEV)
    nat.call("lnn_instnorm_lrelu_bwd", yb, dzb, C + 8, N, V, C, mean, rstd, gamma.detach().to(DEV), beta.detach().to(DEV),
             0.01, dg, db, dbias, 0.5, ws)
    assert rel_err(from_cl_h(yb, C), y.grad) < 3e-3          # dy written in place over y
    assert rel_err(dg.cpu(), 0.5 * gamma.grad) < 1e-3
    assert rel_err(db.cpu(), 0.5 * beta.grad) < 1e-3
    # conv-bias gradient = sum_v dy (~0 analytically after IN backward; only fp16 rounding remains)
    assert float(dbias.abs().max()) <= 2e-3 * float(y.grad.abs().sum((0, 2, 3, 4)).max())


@pytest.mark.parametrize("N,C,K,V3", [(2, 32, 3, (8, 16, 8)), (1, 8, 2, (5, 9, 11)), (1, 320, 3, (3, 4, 3)), (1, 64, 5, (4, 8, 8))])
def test_seg1x1_fwd_bwd(N, C, K, V3):
    z = _rand((N, C) + V3, 1).requires_grad_(True)
    w = (torch.randn((K, C, 1, 1, 1), generator=torch.Generator().manual_seed(2)) * 0.2).requires_grad_(True)
    ref = F.conv3d(z, w)
    dl = torch.randn(ref.shape, generator=torch.Generator().manual_seed(3))
    ref.backward(dl)
    V = V3[0] * V3[1] * V3[2]
    zb, _ = to_cl_h(z.detach())
    wd = w.detach().view(K, C).contiguous().to(DEV)
    logits = torch.zeros((N, K) + V3, device=DEV)
    nat.call("lnn_seg1x1_fwd", zb, C, wd, logits, N, V, C, K)
    assert rel_err(logits.cpu(), ref.detach()) < 1e-5
    base = _rand((N, C) + V3, 7)
    dzb, _ = to_cl_h(base)
    dw = torch.zeros((K, C), device=DEV)
    dw2, dzb2 = dw.clone(), dzb.clone()
    nat.call("lnn_seg1x1_bwd", zb, C, wd, dl.to(DEV), dzb, C, dw, N, V, C, K, 1, 2.0, None)
    ws = torch.empty(nat.query("lnn_seg1x1_bwd_ws_floats", N, C), device=DEV)
    nat.call("lnn_seg1x1_bwd", zb, C, wd, dl.to(DEV), dzb2, C, dw2, N, V, C, K, 1, 2.0, ws)      # workspace path
    assert torch.allclose(dw2, dw, rtol=1e-5, atol=1e-6) and torch.equal(dzb2, dzb)
    assert rel_err(from_cl_h(dzb, C), z.grad + base) < 3e-3
    assert rel_err(dw.cpu(), 2.0 * w.grad.view(K, C)) < 1e-4


@pytest.mark.parametrize("N,C,K,V3", [(2, 32, 3, (24, 16, 20)), (1, 64, 2, (12, 10, 18)), (1, 32, 3, (5, 9, 11)), (2, 8, 3, (4, 8, 8)),
                                      (1, 128, 5, (8, 8, 12)), (1, 256, 3, (4, 6, 4))])
def test_instnorm_lrelu_seg_fwd_fused(N, C, K, V3):
    """lnn_instnorm_lrelu_seg_fwd == lnn_instnorm_lrelu_fwd + lnn_seg1x1_fwd: z bit for bit, logits to fp32 summation order.
    Shapes cover the LDS-staged logits stores (C >= 32, V % 4 == 0: several block trips plus a tail), the direct stores
    (C = 8; V % 4 != 0) and K up to 5."""
    y = q16(_rand((N, C) + V3, 1) * 2 + 0.5)
    g = torch.Generator().manual_seed(9)
    gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).to(DEV), (0.1 * torch.randn(C, generator=g)).to(DEV)
    w = (torch.randn((K, C), generator=g) * 0.2).to(DEV)
    V = V3[0] * V3[1] * V3[2]
    yb, _ = to_cl_h(y)
    mean = torch.empty(N * C, device=DEV); rstd = torch.empty(N * C, device=DEV)
    ws = torch.zeros(nat.query("lnn_instnorm_ws_doubles", N, C), dtype=torch.float64, device=DEV)
    nat.call("lnn_instnorm_stats", yb, N, V, C, 1e-5, mean, rstd, ws)
    z1 = torch.zeros((N,) + V3 + (C + 8,), dtype=torch.float16, device=DEV); z2 = torch.zeros_like(z1)
    l1 = torch.full((N, K) + V3, 7.0, device=DEV); l2 = torch.full((N, K) + V3, 9.0, device=DEV)
    nat.call("lnn_instnorm_lrelu_fwd", yb, z1, C + 8, N, V, C, mean, rstd, gamma, beta, 0.01)
    nat.call("lnn_seg1x1_fwd", z1, C + 8, w, l1, N, V, C, K)
    nat.call("lnn_instnorm_lrelu_seg_fwd", yb, z2, C + 8, N, V, C, mean, rstd, gamma, beta, 0.01, w, l2, K)
    assert torch.equal(z1, z2)
    assert float((l1 - l2).abs().max()) <= 2e-6 * float(l1.abs().max())
    # z = NULL (the last decoder block of a training step: nobody reads its normalised tensor): the same logits, nothing else written
    l3 = torch.full((N, K) + V3, 5.0, device=DEV)
    nat.call("lnn_instnorm_lrelu_seg_fwd", yb, None, C + 8, N, V, C, mean, rstd, gamma, beta, 0.01, w, l3, K)
    assert torch.equal(l3, l2)


@pytest.mark.parametrize("N,K,D,H,W", [(2, 32, 9, 13, 17), (1, 32, 16, 16, 16), (3, 64, 5, 8, 9)])
def test_first_layer_wgrad_with_folded_norm_backward(N, K, D, H, W):
    """lnn_instnorm_lrelu_bwd_sums + lnn_conv3d_wgrad_c1_in_bwd (dy of the first block rebuilt inside its weight gradient) against
    autograd through leaky_relu(instance_norm(conv3d(x))) and against the unfused pair lnn_instnorm_lrelu_bwd + lnn_conv3d_wgrad;
    y must stay untouched.  Ragged tiles, several samples per block range, 2 output-channel blocks."""
    x = _rand((N, 1, D, H, W), 1)
    w = (_rand((K, 1, 3, 3, 3), 2, 0.3)).requires_grad_(True)
    g = torch.Generator().manual_seed(9)
    gamma = (1 + 0.2 * torch.randn(K, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(K, generator=g)).requires_grad_(True)
    V = D * H * W
    xb = x.half().to(DEV).reshape(N, D, H, W).contiguous()
    wp = pack(w.detach().to(DEV), 1, K, 27, 27, 1, 0)
    yb = torch.zeros((N, D, H, W, K), dtype=torch.float16, device=DEV)
    nat.call("lnn_conv3d_fwd", xb, 1, wp, None, yb, K, N, D, H, W, 1, K, 1)
    y = from_cl_h(yb, K).requires_grad_(True)                  # the reference continues from the fp16 y the kernel stored
    z = F.leaky_relu(F.instance_norm(y, weight=gamma, bias=beta, eps=1e-5), 0.01)
    dz = _rand(z.shape, 4)
    z.backward(dz)
    yconv = F.conv3d(x, w, None, padding=1)
    yconv.backward(y.grad)                                     # dL/dw from the reference dy
    mean = torch.empty(N * K, device=DEV); rstd = torch.empty(N * K, device=DEV)
    ws = torch.zeros(nat.query("lnn_instnorm_ws_doubles", N, K), dtype=torch.float64, device=DEV)
    nat.call("lnn_instnorm_stats", yb, N, V, K, 1e-5, mean, rstd, ws)
    ga, be = gamma.detach().to(DEV), beta.detach().to(DEV)
    dzb, _ = to_cl_h(dz, ld=K + 8)
    # unfused pair
    y_u = yb.clone()
    dg_u = torch.zeros(K, device=DEV); db_u = torch.zeros(K, device=DEV)
    nat.call("lnn_instnorm_lrelu_bwd", y_u, dzb, K + 8, N, V, K, mean, rstd, ga, be, 0.01, dg_u, db_u, None, 1.0, ws)
    p_u = torch.zeros(nat.query("lnn_wgrad_panel_elems", 1, K, 27), device=DEV)
    nat.call("lnn_conv3d_wgrad", xb, 1, y_u, K, p_u, N, D, H, W, 1, K, 1)
    # folded
    y_f = yb.clone()
    dg = torch.zeros(K, device=DEV); db = torch.zeros(K, device=DEV)
    nat.call("lnn_instnorm_lrelu_bwd_sums", y_f, dzb, K + 8, N, V, K, mean, rstd, ga, be, 0.01, dg, db, 1.0, ws)
    pf = torch.zeros_like(p_u)
    nat.call("lnn_conv3d_wgrad_c1_in_bwd", xb, y_f, dzb, K + 8, pf, N, D, H, W, K, mean, rstd, ga, be, 0.01, ws, None, 0)
    assert torch.equal(y_f, yb)
    if V > nat.query("lnn_instnorm_small_volume"):
        assert torch.equal(dg, dg_u) and torch.equal(db, db_u)
    else:       # lnn_instnorm_lrelu_bwd is ONE launch on such volumes (csrc/norm_act.hip in_small_bwd_kernel): another summation order
        assert rel_err(dg.cpu(), dg_u.cpu()) < 1e-5 and rel_err(db.cpu(), db_u.cpu()) < 1e-5
    dw_u = torch.zeros((K, 1, 3, 3, 3), device=DEV); dw_f = torch.zeros_like(dw_u)
    nat.call("lnn_unpack_wgrad", p_u, dw_u, 1, K, 27, 27, 1, 0, 1.0, 0)
    nat.call("lnn_unpack_wgrad", pf, dw_f, 1, K, 27, 27, 1, 0, 1.0, 0)
    assert rel_err(dw_f.cpu(), w.grad) < 2e-3
    assert rel_err(dw_f.cpu(), dw_u.cpu()) < 1e-3
    # deterministic mode: ordered reduction of per-writer panel copies, two calls bit-identical
    det = torch.empty(8 * 1024 * 1024, device=DEV)
    outs = []
    for _ in range(2):
        pd = torch.zeros_like(p_u)
        nat.call("lnn_conv3d_wgrad_c1_in_bwd", xb, y_f, dzb, K + 8, pd, N, D, H, W, K, mean, rstd, ga, be, 0.01, ws, det, det.numel())
        outs.append(pd)
    assert torch.equal(outs[0], outs[1]) and rel_err(outs[0].cpu(), pf.cpu()) < 1e-5


@pytest.mark.parametrize("prior", [False, True])
@pytest.mark.parametrize("N,C,K,V3", [(2, 32, 3, (8, 16, 8)), (1, 8, 2, (5, 9, 11)), (2, 320, 3, (3, 4, 3)), (1, 64, 4, (7, 8, 8))])
def test_instnorm_lrelu_seg_bwd_fused(N, C, K, V3, prior):
    """lnn_instnorm_lrelu_seg_bwd (dL/dz of a head-feeding decoder block never written to memory) against autograd through
    conv1x1(leaky_relu(instance_norm(y))) + <prior, z>, and against the two unfused entries it replaces."""
    y = (q16(_rand((N, C) + V3, 1) * 2 + 0.5)).requires_grad_(True)
    g = torch.Generator().manual_seed(9)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    w = (torch.randn((K, C, 1, 1, 1), generator=g) * 0.2).requires_grad_(True)
    V = V3[0] * V3[1] * V3[2]
    yb, _ = to_cl_h(y.detach())
    mean = torch.empty(N * C, device=DEV); rstd = torch.empty(N * C, device=DEV)
    ws = torch.zeros(max(nat.query("lnn_instnorm_lrelu_seg_bwd_ws_doubles", N, C), nat.query("lnn_instnorm_ws_doubles", N, C)),
                     dtype=torch.float64, device=DEV)
    nat.call("lnn_instnorm_stats", yb, N, V, C, 1e-5, mean, rstd, ws)
    ga, be = gamma.detach().to(DEV), beta.detach().to(DEV)
    zb = torch.zeros((N,) + V3 + (C,), dtype=torch.float16, device=DEV)
    nat.call("lnn_instnorm_lrelu_fwd", yb, zb, C, N, V, C, mean, rstd, ga, be, 0.01)
    # reference: the head reads the fp16-rounded z the forward kernel stored (straight-through for the rounding)
    z = F.leaky_relu(F.instance_norm(y, weight=gamma, bias=beta, eps=1e-5), 0.01)
    zq = z + (from_cl_h(zb, C) - z).detach()
    logits = F.conv3d(zq, w)
    dl = torch.randn(logits.shape, generator=g)
    pr = _rand(z.shape, 4) if prior else None
    loss = (logits * dl).sum() + ((z * pr).sum() if prior else 0)
    loss.backward()
    wd = w.detach().view(K, C).contiguous().to(DEV)
    prb = to_cl_h(pr, ld=C + 8)[0] if prior else None
    # unfused pair
    dzb = prb.clone() if prior else torch.zeros((N,) + V3 + (C + 8,), dtype=torch.float16, device=DEV)
    yb_u = yb.clone()
    dw_u = torch.zeros((K, C), device=DEV); dg_u = torch.zeros(C, device=DEV); db_u = torch.zeros(C, device=DEV)
    nat.call("lnn_seg1x1_bwd", zb, C, wd, dl.to(DEV), dzb, C + 8, dw_u, N, V, C, K, 1 if prior else 0, 0.5, None)
    nat.call("lnn_instnorm_lrelu_bwd", yb_u, dzb, C + 8, N, V, C, mean, rstd, ga, be, 0.01, dg_u, db_u, None, 0.5, ws)
    # fused
    dw = torch.zeros((K, C), device=DEV); dg = torch.zeros(C, device=DEV); db = torch.zeros(C, device=DEV)
    nat.call("lnn_instnorm_lrelu_seg_bwd", yb, prb, C + 8, wd, dl.to(DEV), dw, K, N, V, C, mean, rstd, ga, be, 0.01, dg, db, 0.5, ws)
    assert rel_err(from_cl_h(yb, C), y.grad) < 3e-3
    assert rel_err(dg.cpu(), 0.5 * gamma.grad) < 1e-3
    assert rel_err(db.cpu(), 0.5 * beta.grad) < 1e-3
    assert rel_err(dw.cpu(), 0.5 * w.grad.view(K, C)) < 1e-3
    # same rounding points as the unfused pair: agreement to fp32 summation order (dy differs by at most an fp16 ulp where it does)
    assert rel_err(from_cl_h(yb, C), from_cl_h(yb_u, C)) < 1e-3
    # (the fused reduction keeps dz and z in fp32 where the unfused pair rounds both to fp16 in memory: 2^-11 noise per element)
    assert rel_err(dw.cpu(), dw_u.cpu()) < 1e-3 and rel_err(dg.cpu(), dg_u.cpu()) < 1e-3 and rel_err(db.cpu(), db_u.cpu()) < 1e-3


@pytest.mark.parametrize("batch_dice", [0, 1])
@pytest.mark.parametrize("N,K,V3", [(2, 3, (8, 16, 8)), (3, 2, (5, 9, 11)), (1, 5, (4, 8, 8))])
def test_dice_ce_fwd_bwd(N, K, V3, batch_dice):
    from oracle import losses
    g = torch.Generator().manual_seed(1)
    logits = (torch.randn((N, K) + V3, generator=g) * 2).requires_grad_(True)
    labels = torch.randint(0, K, (N, 1) + V3, generator=g).float()
    if K > 2:
        labels[0][labels[0] == K - 1] = 0      # an empty foreground class in sample 0
    ref = losses.dc_and_ce_loss(logits, labels, bool(batch_dice))
    ref.backward()
    V = V3[0] * V3[1] * V3[2]
    lg, lb = logits.detach().to(DEV), labels.to(DEV)
    ws = torch.zeros(nat.query("lnn_dice_ce_ws_doubles", N, K), dtype=torch.float64, device=DEV)
    out = torch.zeros(1, device=DEV)
    nat.call("lnn_dice_ce_fwd", lg, lb, N, K, V, batch_dice, 1e-5, out, ws)
    assert abs(float(out) - float(ref)) <= 1e-5 * abs(float(ref))       # north_star: 1e-4 relative loss
    dl = torch.zeros_like(lg)
    nat.call("lnn_dice_ce_bwd", lg, lb, N, K, V, batch_dice, 1e-5, ws, 1.5, torch.full((1,), 2.0, device=DEV), 1.0, dl)
    assert rel_err(dl.cpu(), 3.0 * logits.grad) < 1e-4


@pytest.mark.parametrize("K", [2, 3, 5])
def test_dice_ce_with_infinite_and_very_negative_logits(K):
    """A logit of -inf (or -1e30) is a class of probability exactly 0: torch's softmax / log_softmax give a finite loss and a zero
    gradient for that channel as long as the voxel's label is another class; the kernels' compensated hardware exp must do the same
    (its residual term was inf - inf = NaN before round 5)."""
    from oracle import losses
    N, V3 = 2, (4, 8, 8)
    g = torch.Generator().manual_seed(5)
    logits = torch.randn((N, K) + V3, generator=g) * 2
    labels = torch.randint(0, K, (N, 1) + V3, generator=g).float()
    lab = labels[:, 0].long()
    kill = torch.zeros_like(logits, dtype=torch.bool)
    for k in range(K):                       # every 7th voxel: one channel that is NOT the label goes to -inf, the next 7th to -1e30
        idx = ((torch.arange(lab.numel()).reshape(lab.shape) % 7) == 0) & (lab != k) & ~kill.any(1)
        kill[:, k] |= idx
    big = torch.zeros_like(kill)
    big[:, 0] = ((torch.arange(lab.numel()).reshape(lab.shape) % 7) == 3) & (lab != 0) & ~kill.any(1)
    logits[kill] = float("-inf")
    logits[big] = -1e30
    lg = logits.clone().requires_grad_(True)
    ref = losses.dc_and_ce_loss(lg, labels, False)
    ref.backward()
    assert torch.isfinite(ref) and torch.isfinite(lg.grad).all()
    V = V3[0] * V3[1] * V3[2]
    dg, lb = logits.to(DEV), labels.to(DEV)
    ws = torch.zeros(nat.query("lnn_dice_ce_ws_doubles", N, K), dtype=torch.float64, device=DEV)
    out = torch.zeros(1, device=DEV)
    nat.call("lnn_dice_ce_fwd", dg, lb, N, K, V, 0, 1e-5, out, ws)
    assert abs(float(out) - float(ref)) <= 1e-5 * abs(float(ref))
    dl = torch.full_like(dg, float("nan"))
    nat.call("lnn_dice_ce_bwd", dg, lb, N, K, V, 0, 1e-5, ws, 1.0, torch.full((1,), 1.0, device=DEV), 1.0, dl)
    assert torch.isfinite(dl).all() and rel_err(dl.cpu(), lg.grad) < 1e-4
    assert float(dl.cpu()[kill].abs().max()) == 0.0


@pytest.mark.parametrize("N,C,K,V3", [(2, 320, 3, (10, 12, 10)), (2, 320, 3, (5, 6, 5)), (1, 256, 3, (7, 5, 3)), (3, 32, 2, (9, 11, 5)),
                                      (1, 8, 5, (4, 8, 8)), (2, 48, 4, (3, 7, 5)), (1, 512, 1, (2, 3, 5)), (1, 640, 3, (4, 4, 4)),
                                      (2, 24, 8, (5, 5, 5))])
def test_seg1x1_fwd_wave_kernel_shapes(N, C, K, V3):
    """The 1x1x1 head forward with thread = (voxel, channel octet) (C <= 512; round 5): channel counts whose octet count is not a
    power of two (320 -> 40 of 64 lanes, 48 -> 6 of 8, 24 -> 3 of 4), one octet (C = 8), runtime K > 4, ragged voxel counts, a
    channel stride wider than C, and the voxel-per-thread kernel that still serves C > 512."""
    z = _rand((N, C) + V3, 11)
    w = torch.randn((K, C, 1, 1, 1), generator=torch.Generator().manual_seed(2)) * 0.2
    ref = F.conv3d(q16(z), w)
    V = V3[0] * V3[1] * V3[2]
    zb, _ = to_cl_h(z)
    wide = torch.zeros((N,) + V3 + (C + 16,), dtype=torch.float16, device=DEV)
    wide[..., :C] = zb
    wide[..., C:] = 77.0                                   # must never be read
    wd = w.view(K, C).contiguous().to(DEV)
    for buf, ld in ((zb, C), (wide, C + 16)):
        logits = torch.full((N, K) + V3, 3.0, device=DEV)
        nat.call("lnn_seg1x1_fwd", buf, ld, wd, logits, N, V, C, K)
        assert rel_err(logits.cpu(), ref) < 1e-5


def test_dice_ce_golden(golden_dir):
    d = np.load(golden_dir + "/dice_ce.npz")
    lg = torch.from_numpy(d["logits"]).to(DEV); lb = torch.from_numpy(d["target"]).to(DEV)
    N, K = lg.shape[:2]; V = lg[0, 0].numel()
    ws = torch.zeros(nat.query("lnn_dice_ce_ws_doubles", N, K), dtype=torch.float64, device=DEV)
    out = torch.zeros(1, device=DEV)
    for bd, key in ((0, "loss_sample_dice"), (1, "loss_batch_dice")):
        nat.call("lnn_dice_ce_fwd", lg, lb, N, K, V, bd, 1e-5, out, ws)
        assert abs(float(out) - float(d[key])) <= 1e-5 * abs(float(d[key]))
    counts = torch.zeros((N, K - 1, 3), device=DEV)
    nat.call("lnn_online_dice_counts", lg, lb, N, K, V, counts)
    c = counts.cpu().numpy()
    assert np.array_equal(c[:, :, 0], d["tp"]) and np.array_equal(c[:, :, 1], d["fp"]) and np.array_equal(c[:, :, 2], d["fn"])


def test_kl_logits_golden(golden_dir):
    d = np.load(golden_dir + "/lwf_reference.npz")
    for T in (1, 2):
        for i in range(2):
            p = torch.from_numpy(d[f"pred_{i}"]).to(DEV); t = torch.from_numpy(d[f"teach_{i}"]).to(DEV)
            N, K = p.shape[:2]; V = p[0, 0].numel()
            out = torch.zeros(1, device=DEV)
            ws = torch.full((nat.query("lnn_kl_logits_ws_doubles", N),), float("nan"), dtype=torch.float64, device=DEV)    # contents must not matter
            nat.call("lnn_kl_logits", p, t, N, K, V, float(T), out, ws)
            exp = float(d[f"kl{i}_T{T}"])
            assert abs(float(out) - exp) <= 1e-5 * abs(exp)


def test_param_kernels():
    n = 1_000_003
    g = torch.Generator().manual_seed(1)
    th, ts, f, gr = (torch.randn(n, generator=g) for _ in range(4))
    f = f.abs()
    thd, tsd, fd, grd = th.to(DEV), ts.to(DEV), f.to(DEV), gr.to(DEV)
    out = torch.zeros(1, device=DEV); ws = torch.zeros(nat.query("lnn_flat_reduce_ws_doubles"), dtype=torch.float64, device=DEV)
    nat.call("lnn_ewc_penalty_fwd", thd, tsd, fd, n, 0.4, out, ws)
    exp = 0.2 * (f.double() * (th.double() - ts.double()) ** 2).sum()
    assert abs(float(out) - float(exp)) <= 1e-6 * float(exp)
    g2 = grd.clone()
    nat.call("lnn_ewc_penalty_bwd", thd, tsd, fd, n, 0.4, 1.0, torch.full((1,), 2.0, device=DEV), g2)
    assert rel_err(g2.cpu(), gr + 2.0 * 0.4 * f * (th - ts)) < 1e-6
    fo = torch.zeros(n, device=DEV)
    nat.call("lnn_fisher_square", grd, fo, n, 0.5)
    assert rel_err(fo.cpu(), (0.5 * gr) ** 2) < 1e-6
    nat.call("lnn_fisher_accumulate", grd, fo, n, 1.0, 0.25)
    assert rel_err(fo.cpu(), (0.5 * gr) ** 2 + 0.25 * gr ** 2) < 1e-6
    nat.call("lnn_fisher_ema", grd, fo, n, 1.0, 0.1)
    assert rel_err(fo.cpu(), 0.1 * gr ** 2 + 0.9 * ((0.5 * gr) ** 2 + 0.25 * gr ** 2)) < 1e-6
    nat.call("lnn_gradnorm_sumsq", grd, n, 0.5, ws, 1)
    assert abs(float(ws[0]) - float((0.5 * gr.double()).pow(2).sum())) <= 1e-6 * float(ws[0])
    assert float(ws[1]) == 0
    first = ws[:2].clone()
    nat.call("lnn_gradnorm_sumsq", grd, n, 0.5, ws, 1)
    assert torch.equal(first, ws[:2])        # deterministic reduction: bit-identical on repetition
    nat.call("lnn_gradnorm_sumsq", grd[1:], n - 1, 0.5, ws, 0)      # misaligned range, accumulating
    assert abs(float(ws[0]) - float((0.5 * gr.double()).pow(2).sum() + (0.5 * gr[1:].double()).pow(2).sum())) <= 1e-6 * float(ws[0])
    bad = grd.clone(); bad[5] = float("inf"); bad[7] = float("nan")
    nat.call("lnn_gradnorm_sumsq", bad, n, 1.0, ws, 1)
    assert float(ws[1]) == 2
    # two SGD-Nesterov steps against torch.optim.SGD
    p = torch.nn.Parameter(th.clone())
    opt = torch.optim.SGD([p], 1e-2, momentum=0.99, nesterov=True, weight_decay=3e-5)
    buf = torch.zeros(n, device=DEV); thg = thd.clone()
    for step in range(2):
        p.grad = gr * (step + 1)
        opt.step()
        nat.call("lnn_sgd_nesterov_step", thg, buf, grd * (step + 1) * 4.0, n, 1e-2, 0.99, 3e-5, 0.25, 1 if step == 0 else 0)
    assert rel_err(thg.cpu(), p.detach()) < 1e-6
    # device-controlled clip + skip: same two steps with clip_grad_norm_(12) on a scaled gradient
    p = torch.nn.Parameter(th.clone())
    opt = torch.optim.SGD([p], 1e-2, momentum=0.99, nesterov=True, weight_decay=3e-5)
    buf.zero_(); thg = thd.clone()
    for step in range(2):
        p.grad = gr.clone() * (step + 1)
        torch.nn.utils.clip_grad_norm_([p], 12)
        opt.step()
        gs = grd * (step + 1) * 1024.0
        nat.call("lnn_gradnorm_sumsq", gs, n, 1 / 1024.0, ws, 1)
        nat.call("lnn_sgd_nesterov_step_clipped", thg, buf, gs, n, 1e-2, 0.99, 3e-5, 1 / 1024.0, 12.0, ws)
    assert rel_err(thg.cpu(), p.detach()) < 1e-6
    before = thg.clone()
    nat.call("lnn_gradnorm_sumsq", bad, n, 1.0, ws, 1)
    nat.call("lnn_sgd_nesterov_step_clipped", thg, buf, bad, n, 1e-2, 0.99, 3e-5, 1.0, 12.0, ws)
    assert torch.equal(before, thg)          # non-finite gradient -> step skipped


def test_batched_pack_unpack_match_per_layer():
    """lnn_pack_weights_batched / lnn_unpack_wgrad_batched == the per-layer entry points, bit for bit."""
    layers = [(27, 32, 16, 16 * 27, 27, 1), (27, 16, 32, 27, 16 * 27, 1), (8, 24, 40, 8, 24 * 8, 1), (1, 32, 27, 27, 1, 0)]
    g = torch.Generator().manual_seed(5)
    src = torch.randn(40000, generator=g).to(DEV)
    desc, soff, doff, first = [], 0, 0, 0
    per_layer = []
    for (nt, M, KC, sm, skc, st) in layers:
        n = nat.query("lnn_packed_weight_elems", nt, M, KC)
        ref = torch.zeros(n, dtype=torch.float16, device=DEV)
        nat.call("lnn_pack_weights", src[soff:], ref, nt, M, KC, sm, skc, st)
        per_layer.append(ref)
        desc.append([soff, doff, sm, skc, st, nt, M, KC, first])
        soff += nt * M * KC; doff += n; first += n
    out = torch.zeros(doff, dtype=torch.float16, device=DEV)
    d = torch.tensor(desc, dtype=torch.int64, device=DEV)
    nat.call("lnn_pack_weights_batched", src, out, d, len(layers), first)
    assert torch.equal(out, torch.cat(per_layer))
    # unpack
    desc, poff, goff, first = [], 0, 0, 0
    panels, refs = [], []
    for (nt, M, KC, sm, skc, st) in layers:
        n = nat.query("lnn_wgrad_panel_elems", nt, M, KC)
        pan = torch.randn(n, generator=g).to(DEV)
        ref = torch.full((nt * M * KC,), 0.5, device=DEV)
        nat.call("lnn_unpack_wgrad", pan, ref, nt, M, KC, sm, skc, st, 2.0, 1)
        panels.append(pan); refs.append(ref)
        desc.append([poff, goff, sm, skc, st, nt, M, KC, first])
        poff += n; goff += nt * M * KC; first += nt * M * KC
    grad = torch.full((goff,), 0.5, device=DEV)
    nat.call("lnn_unpack_wgrad_batched", torch.cat(panels), grad, torch.tensor(desc, dtype=torch.int64, device=DEV),
             len(layers), first, 2.0, 1)
    assert torch.equal(grad, torch.cat(refs))


def test_batch_dice_data_parallel_exchange_equals_full_batch():
    """batch Dice under data parallelism = all-reduce of tp/fp/fn inside the loss.  Two "ranks" are
    emulated in one process (the all-reduce is the sum of the two workspaces' totals): mean of the rank losses ==
    full-batch loss, and rank gradient / world == the full-batch gradient of the same samples."""
    N, K, V, W = 4, 3, 6 * 8 * 6, 2
    g = torch.Generator().manual_seed(21)
    logits = (torch.randn((N, K, V), generator=g) * 2).to(DEV)
    labels = torch.randint(0, K, (N, 1, V), generator=g).float().to(DEV)
    nws = nat.query("lnn_dice_ce_ws_doubles", N, K)
    ws, out = torch.zeros(nws, dtype=torch.float64, device=DEV), torch.zeros(1, device=DEV)
    nat.call("lnn_dice_ce_fwd", logits, labels, N, K, V, 1, 1e-5, out, ws)
    full_loss = float(out)
    dfull = torch.empty_like(logits)
    nat.call("lnn_dice_ce_bwd", logits, labels, N, K, V, 1, 1e-5, ws, 1.0, None, 1.0, dfull)
    Nr = N // W
    wss, lgs, lbs = [], [], []
    for r in range(W):
        lgs.append(logits[r * Nr:(r + 1) * Nr].contiguous()); lbs.append(labels[r * Nr:(r + 1) * Nr].contiguous())
        w_r = torch.zeros(nat.query("lnn_dice_ce_ws_doubles", Nr, K), dtype=torch.float64, device=DEV)
        nat.call("lnn_dice_ce_fwd", lgs[r], lbs[r], Nr, K, V, 1, 1e-5, out, w_r)
        wss.append(w_r)
    tot = wss[0][:Nr * K * 3] + wss[1][:Nr * K * 3]                    # dist.all_reduce(ws[:N*K*3], SUM)
    losses, grads = [], []
    for r in range(W):
        wss[r][:Nr * K * 3] = tot
        nat.call("lnn_dice_ce_loss_from_totals", wss[r], Nr, K, V, 1, 1e-5, out)
        losses.append(float(out))
        d = torch.empty_like(lgs[r])
        nat.call("lnn_dice_ce_bwd", lgs[r], lbs[r], Nr, K, V, 1, 1e-5, wss[r], 1.0, None, float(W), d)
        grads.append(d / W)                                            # the 1/world of the gradient average
    assert abs(sum(losses) / W - full_loss) <= 1e-6 * abs(full_loss)
    got = torch.cat(grads)
    assert float((got - dfull).abs().max()) <= 1e-6 * float(dfull.abs().max())


@pytest.mark.parametrize("which", [5, 9, 10])
def test_every_stride1_conv_kernel_variant(which):
    """The automatic selection picks the tile kernel (v5) / the z-streaming kernel (v9) / the macro-tile kernel by layer shape, so the
    small parity shapes above only exercise v5: pin each shipped stride-1 kernel in turn and run forward + dgrad (with and without accumulation) on all
    stride-1 cases, including ragged extents and output channels that are not a multiple of 64.  The 32- / 64-channel
    cases cover the four wave-role configurations of v9 (chunks x output blocks x footprints), forward and dgrad."""
    cases = [c for c in CONV_CASES if c[6] == 1] + [(1, 32, 96, 9, 8, 17, 1), (2, 128, 64, 8, 8, 8, 1), (1, 24, 160, 5, 6, 7, 1),
                                                    (1, 32, 64, 7, 9, 18, 1), (1, 64, 64, 6, 10, 9, 1), (2, 64, 128, 5, 12, 9, 1),
                                                    (1, 32, 32, 37, 5, 21, 1),
                                                    # 128 input channels: v9 with all eight waves as channel chunks (round 3)
                                                    (1, 128, 128, 7, 9, 18, 1), (1, 128, 96, 5, 6, 10, 1), (1, 128, 256, 4, 5, 9, 1),
                                                    # last z tile with 4 / 3 / 4 live planes (the tile kernel's half steps: two waves
                                                    # per live plane, round 5) behind one or two full tiles, and 20 = 8 + 8 + 4
                                                    (1, 48, 40, 20, 11, 9, 1), (1, 32, 32, 11, 8, 16, 1), (2, 16, 32, 12, 8, 8, 1)]
    assert nat.lib().lnn_debug_force_conv_kernel(which) == 0
    try:
        for (N, C, K, D, H, W, s) in cases:
            test_conv3d_fwd(N, C, K, D, H, W, s)
            for acc in (0, 1):
                test_conv3d_dgrad(N, C, K, D, H, W, s, acc)
    finally:
        nat.lib().lnn_debug_force_conv_kernel(-1)


@pytest.mark.parametrize("N,C,K,D,H,W,cat", [(2, 32, 32, 9, 8, 17, 0), (1, 64, 32, 6, 10, 9, 1), (2, 64, 64, 5, 12, 9, 0),
                                              (1, 16, 32, 5, 9, 11, 0), (1, 32, 32, 37, 5, 21, 0), (2, 128, 64, 34, 6, 10, 0),
                                              (1, 128, 128, 33, 9, 9, 0), (2, 1, 32, 9, 10, 17, 0), (1, 1, 32, 4, 8, 8, 0)])
def test_conv3d_fwd_in_stats(N, C, K, D, H, W, cat):
    """lnn_conv3d_fwd_in_stats == lnn_conv3d_fwd followed by lnn_instnorm_stats: the output bit for bit, mean / rstd to
    fp32 summation order (the z-streaming kernel takes the sums in its epilogue from the values it stores; for shapes it
    does not cover the entry runs the two passes itself), and both equal the statistics of the stored fp16 tensor."""
    x = _rand((N, C, D, H, W), 1)
    w = _rand((K, C, 3, 3, 3), 2, 0.1)
    b = torch.randn(K, generator=torch.Generator().manual_seed(3)).to(DEV)
    xb, _ = to_cl_h(x)
    wp = pack(w.to(DEV), 1, K, 27, 27, 1, 0) if C == 1 else pack_conv_fwd(w.to(DEV))    # first layer: taps are the contraction
    V = D * H * W
    ws = torch.zeros(nat.query("lnn_instnorm_ws_doubles", N, K), dtype=torch.float64, device=DEV)
    y1 = torch.zeros((N, D, H, W, K), dtype=torch.float16, device=DEV); y2 = torch.zeros_like(y1)
    m1, r1, m2, r2 = (torch.zeros(N * K, device=DEV) for _ in range(4))
    nat.call("lnn_conv3d_fwd", xb, C, wp, b, y1, K, N, D, H, W, C, K, 1)
    nat.call("lnn_instnorm_stats", y1, N, V, K, 1e-5, m1, r1, ws)
    if cat:
        xa = xb[..., :C // 2].contiguous(); xc = xb[..., C // 2:].contiguous()
        nat.call("lnn_conv3d_fwd_in_stats", xa, xc, C // 2, C // 2, wp, b, y2, N, D, H, W, C, K, 1, 1e-5, m2, r2, ws, None, 0)
    else:
        nat.call("lnn_conv3d_fwd_in_stats", xb, None, C, 0, wp, b, y2, N, D, H, W, C, K, 1, 1e-5, m2, r2, ws, None, 0)
    assert torch.equal(y1, y2)
    yf = y1.float().reshape(N, V, K)
    mean = yf.mean(1).reshape(-1); var = yf.var(1, unbiased=False).reshape(-1)
    for m, r in ((m1, r1), (m2, r2)):
        assert float((m - mean).abs().max()) <= 1e-5 * float(mean.abs().max()) + 1e-6
        assert float((r - (var + 1e-5).rsqrt()).abs().max()) <= 1e-5 * float(r.abs().max())
    assert float((m1 - m2).abs().max()) <= 2e-6 * float(m1.abs().max()) + 1e-7 and float((r1 - r2).abs().max()) <= 2e-6 * float(r1.abs().max())


@pytest.mark.parametrize("N,C,K,D,H,W", [(2, 32, 64, 16, 32, 16), (1, 32, 128, 34, 20, 36), (1, 64, 64, 12, 20, 36),
                                         (1, 64, 128, 36, 12, 20), (1, 32, 64, 6, 18, 34)])
def test_stride2_conv_streaming_kernel(N, C, K, D, H, W):
    """igemm_down2s (z-streaming stride-2 conv forward, round 3) pinned by lnn_debug_force_down2_kernel(1): against
    F.conv3d(stride=2, padding=1), against the tile kernel it replaces, and its fused InstanceNorm statistics against the
    separate pass (output bit for bit, mean / rstd to fp32 summation order).  Ragged footprints (Ho, Wo not multiples of the
    8 x 8 / 4 x 8 block footprint), both wave-role configurations (32 / 64 input channels), 1 and 2 output-channel groups."""
    x = _rand((N, C, D, H, W), 1)
    w = _rand((K, C, 3, 3, 3), 2, 0.1)
    b = torch.randn(K, generator=torch.Generator().manual_seed(3))
    ref = F.conv3d(x, w, b, stride=2, padding=1)
    xb, _ = to_cl_h(x)
    wp = pack_conv_fwd(w.to(DEV))
    Do, Ho, Wo = ref.shape[2:]
    V = Do * Ho * Wo
    outs = {}
    try:
        for which in (0, 1):
            assert nat.lib().lnn_debug_force_down2_kernel(which) == 0
            yb = torch.full((N, Do, Ho, Wo, K), 7.0, dtype=torch.float16, device=DEV)
            nat.call("lnn_conv3d_fwd", xb, C, wp, b.to(DEV), yb, K, N, D, H, W, C, K, 2)
            outs[which] = yb
            assert rel_err(from_cl_h(yb, K), ref) < 2e-3, which
        assert rel_err(from_cl_h(outs[1], K), from_cl_h(outs[0], K)) < 1e-3
        # output into a wider buffer at a channel offset (the engine never does this for a strided conv, the C-ABI allows it)
        yw = torch.full((N, Do, Ho, Wo, K + 16), 7.0, dtype=torch.float16, device=DEV)
        nat.call("lnn_conv3d_fwd", xb, C, wp, b.to(DEV), View(yw, 16), K + 16, N, D, H, W, C, K, 2)
        assert torch.equal(yw[..., 16:], outs[1]) and bool((yw[..., :16] == 7.0).all())
        # fused statistics
        ws = torch.zeros(nat.query("lnn_instnorm_ws_doubles", N, K), dtype=torch.float64, device=DEV)
        m1, r1, m2, r2 = (torch.zeros(N * K, device=DEV) for _ in range(4))
        nat.call("lnn_instnorm_stats", outs[1], N, V, K, 1e-5, m1, r1, ws)
        y2 = torch.zeros_like(outs[1])
        nat.call("lnn_conv3d_fwd_in_stats", xb, None, C, 0, wp, b.to(DEV), y2, N, D, H, W, C, K, 2, 1e-5, m2, r2, ws, None, 0)
        assert torch.equal(y2, outs[1])
        assert float((m1 - m2).abs().max()) <= 2e-6 * float(m1.abs().max()) + 1e-7 and float((r1 - r2).abs().max()) <= 2e-6 * float(r1.abs().max())
    finally:
        nat.lib().lnn_debug_force_down2_kernel(-1)


@pytest.mark.parametrize("N,C,K,D,H,W", [(2, 64, 32, 8, 16, 8), (1, 128, 64, 17, 10, 18), (1, 64, 32, 3, 9, 17), (1, 128, 32, 6, 4, 8)])
def test_convT_dgrad_streaming_kernel(N, C, K, D, H, W):
    """The 2x2x2 variant of igemm_down2s (data gradient of the transposed conv: 8 taps, no padding, 32 / 64 gathered channels
    = K) pinned by lnn_debug_force_down2_kernel(1): against autograd through F.conv_transpose3d and against the tile kernel."""
    x = _rand((N, C, D, H, W), 1).requires_grad_(True)
    w = _rand((C, K, 2, 2, 2), 2, 0.1).requires_grad_(True)
    ref = F.conv_transpose3d(x, w, None, stride=2)
    dy = _rand(ref.shape, 4)
    ref.backward(dy)
    dyb, _ = to_cl_h(dy, ld=K + 8, offset=8)
    wp = pack_convT_dgrad(w.detach().to(DEV))
    outs = {}
    try:
        for which in (0, 1):
            assert nat.lib().lnn_debug_force_down2_kernel(which) == 0
            dxb = torch.full((N, D, H, W, C + 16), 7.0, dtype=torch.float16, device=DEV)
            nat.call("lnn_convT3d_k2s2_dgrad", View(dyb, 8), K + 8, wp, dxb, C + 16, N, D, H, W, C, K, 0)
            assert rel_err(from_cl_h(dxb, C), x.grad) < 3e-3, which
            assert bool((dxb[..., C:] == 7.0).all())
            outs[which] = dxb
        assert rel_err(from_cl_h(outs[1], C), from_cl_h(outs[0], C)) < 1e-3
    finally:
        nat.lib().lnn_debug_force_down2_kernel(-1)


@pytest.mark.parametrize("zseg", [0, 3])
@pytest.mark.parametrize("N,C,D,H,W", [(2, 32, 9, 8, 17), (1, 32, 37, 5, 21), (3, 32, 4, 4, 8), (2, 64, 6, 10, 9), (1, 64, 13, 12, 7),
                                       (1, 128, 5, 6, 10)])
def test_dgrad_with_fused_norm_backward_reduce(N, C, D, H, W, zseg):
    """lnn_conv3d_dgrad_in_bwd_sums == lnn_conv3d_dgrad_ws + lnn_instnorm_lrelu_bwd_sums on the block whose output the convolution
    consumed: dL/dz bit for bit, the (sample, channel) sums and the affine gradients to summation order -- with the z-streaming
    kernel forced the reduce rides its epilogue for 32 -> 32 channels (ragged footprints, z segments, negative gamma); 64 / 128
    channels have no fused instance and run the two calls -- and lnn_instnorm_lrelu_bwd_apply on those sums gives dL/dy of autograd."""
    K = C
    g = torch.Generator().manual_seed(11)
    u = q16(_rand((N, C, D, H, W), 1) * 1.5 + 0.3).requires_grad_(True)           # convolution output of the stage's first block
    gamma = (1 + 0.3 * torch.randn(C, generator=g)); gamma[::5] *= -1.0
    gamma = gamma.requires_grad_(True)
    beta = (0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    w = _rand((K, C, 3, 3, 3), 2, 0.1)
    z = F.leaky_relu(F.instance_norm(u, weight=gamma, bias=beta, eps=1e-5), 0.01)
    z.retain_grad()
    y2 = F.conv3d(z, w, None, padding=1)
    dy = _rand(y2.shape, 4)
    y2.backward(dy)
    V = D * H * W
    ub, _ = to_cl_h(u.detach())
    dyb, _ = to_cl_h(dy)
    wp = pack_conv_dgrad(w.to(DEV))
    mean = torch.empty(N * C, device=DEV); rstd = torch.empty(N * C, device=DEV)
    nws = nat.query("lnn_instnorm_ws_doubles", N, C)
    ws0 = torch.zeros(nws, dtype=torch.float64, device=DEV)
    nat.call("lnn_instnorm_stats", ub, N, V, C, 1e-5, mean, rstd, ws0)
    ga, be = gamma.detach().to(DEV), beta.detach().to(DEV)
    res = []
    try:
        assert nat.lib().lnn_debug_force_conv_kernel(9) == 0
        assert nat.lib().lnn_debug_set_v9_zseg(zseg) == 0
        for fused in (False, True):
            dx = torch.full((N, D, H, W, C + 8), 7.0, dtype=torch.float16, device=DEV)
            ws = torch.zeros(nws, dtype=torch.float64, device=DEV)
            dg = torch.zeros(C, device=DEV); db = torch.zeros(C, device=DEV)
            if fused:
                nat.call("lnn_conv3d_dgrad_in_bwd_sums", dyb, K, wp, dx, C + 8, N, D, H, W, C, K, ub, mean, rstd, ga, be, 0.01, dg, db,
                         0.5, ws, None, 0)
                assert nat.lib().lnn_debug_last_dgrad_reduce_fused() == (1 if C == 32 else 0)
            else:
                nat.call("lnn_conv3d_dgrad_ws", dyb, K, wp, dx, C + 8, N, D, H, W, C, K, 1, 0, None, 0)
                nat.call("lnn_instnorm_lrelu_bwd_sums", ub, dx, C + 8, N, V, C, mean, rstd, ga, be, 0.01, dg, db, 0.5, ws)
            res.append((dx, ws[:N * C * 3].view(N * C, 3)[:, :2].clone(), dg, db, ws))
    finally:
        nat.lib().lnn_debug_set_v9_zseg(0)
        nat.lib().lnn_debug_force_conv_kernel(-1)
    (dx0, s0, dg0, db0, _), (dx1, s1, dg1, db1, ws1) = res
    assert torch.equal(dx0[..., :C], dx1[..., :C]) and bool((dx1[..., C:] == 7.0).all())
    assert rel_err(from_cl_h(dx1, C), z.grad) < 3e-3
    scale = s0.abs().max(0).values
    assert float(((s0 - s1).abs() / scale).max()) < 2e-5, ((s0 - s1).abs() / scale).max(0)
    assert rel_err(dg1.cpu(), dg0.cpu()) < 2e-5 and rel_err(db1.cpu(), db0.cpu()) < 2e-5
    assert rel_err(dg1.cpu(), 0.5 * gamma.grad) < 2e-3 and rel_err(db1.cpu(), 0.5 * beta.grad) < 2e-3
    nat.call("lnn_instnorm_lrelu_bwd_apply", ub, dx1, C + 8, N, V, C, mean, rstd, ga, be, 0.01, ws1)
    assert rel_err(from_cl_h(ub, C), u.grad) < 3e-3             # dy in place over u


def test_fused_norm_backward_reduce_with_a_large_channel_offset():
    """The fused reduce sums g u of the UN-normalised convolution output in fp32 and forms sum g xhat = rstd (sum g u - mean sum g) in
    fp64 afterwards: with |mean| = 30 std per channel the cancellation must stay far below the fp16 rounding of the operands
    (profiles/r05_fused_reduce_conditioning.txt: 3e-6 of the largest sum here, 7e-6 at 100 std)."""
    N, C, D, H, W = 2, 32, 24, 16, 24
    K, V = C, 24 * 16 * 24
    g = torch.Generator().manual_seed(11)
    u = q16(torch.randn((N, C, D, H, W), generator=g) + 30.0)
    gamma, beta = (1 + 0.3 * torch.randn(C, generator=g)).to(DEV), (0.2 * torch.randn(C, generator=g)).to(DEV)
    w = torch.randn((K, C, 3, 3, 3), generator=g) * 0.1
    dy = torch.randn((N, K, D, H, W), generator=g)
    ub, _ = to_cl_h(u)
    dyb, _ = to_cl_h(dy)
    wp = pack_conv_dgrad(w.to(DEV))
    mean = torch.empty(N * C, device=DEV); rstd = torch.empty(N * C, device=DEV)
    nws = nat.query("lnn_instnorm_ws_doubles", N, C)
    nat.call("lnn_instnorm_stats", ub, N, V, C, 1e-5, mean, rstd, torch.zeros(nws, dtype=torch.float64, device=DEV))
    sums = []
    try:
        assert nat.lib().lnn_debug_force_conv_kernel(9) == 0
        for fused in (False, True):
            dx = torch.zeros((N, D, H, W, C), dtype=torch.float16, device=DEV)
            ws = torch.zeros(nws, dtype=torch.float64, device=DEV)
            dg = torch.zeros(C, device=DEV); db = torch.zeros(C, device=DEV)
            if fused:
                nat.call("lnn_conv3d_dgrad_in_bwd_sums", dyb, K, wp, dx, C, N, D, H, W, C, K, ub, mean, rstd, gamma, beta, 0.01, dg, db, 1.0,
                         ws, None, 0)
                assert nat.lib().lnn_debug_last_dgrad_reduce_fused() == 1
            else:
                nat.call("lnn_conv3d_dgrad_ws", dyb, K, wp, dx, C, N, D, H, W, C, K, 1, 0, None, 0)
                nat.call("lnn_instnorm_lrelu_bwd_sums", ub, dx, C, N, V, C, mean, rstd, gamma, beta, 0.01, dg, db, 1.0, ws)
            sums.append(ws[:N * C * 3].view(N * C, 3)[:, :2].cpu().clone())
    finally:
        nat.lib().lnn_debug_force_conv_kernel(-1)
    s0, s1 = sums
    for j in range(2):
        assert float((s1[:, j] - s0[:, j]).abs().max()) <= 2e-5 * float(s0[:, j].abs().max())


@pytest.mark.parametrize("C", [32, 128])
@pytest.mark.parametrize("zseg", [2, 3, 5])
def test_v9_z_segments(zseg, C):
    """v9 walks a column of output planes; long columns can be cut into z segments (item = column x segment): every cut
    must reproduce the unsegmented result bit for bit (each segment re-reads one halo plane on either side)."""
    N, K, D, H, W = 1, 32, 23, 9, 17
    x = _rand((N, C, D, H, W), 1)
    w = _rand((K, C, 3, 3, 3), 2, 0.1)
    b = torch.randn(K, generator=torch.Generator().manual_seed(3))
    ref = F.conv3d(x, w, b, padding=1)
    xb, _ = to_cl_h(x)
    wp = pack_conv_fwd(w.to(DEV))
    outs = []
    try:
        assert nat.lib().lnn_debug_force_conv_kernel(9) == 0
        for z in (1, zseg):
            assert nat.lib().lnn_debug_set_v9_zseg(z) == 0
            yb = torch.zeros((N, D, H, W, K), dtype=torch.float16, device=DEV)
            nat.call("lnn_conv3d_fwd", xb, C, wp, b.to(DEV), yb, K, N, D, H, W, C, K, 1)
            outs.append(yb)
    finally:
        nat.lib().lnn_debug_set_v9_zseg(0)
        nat.lib().lnn_debug_force_conv_kernel(-1)
    assert rel_err(from_cl_h(outs[0], K), ref) < 2e-3
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("N,Ca,Cb,K,D,H,W", [(2, 32, 32, 32, 8, 16, 8), (1, 32, 32, 64, 9, 8, 17), (1, 64, 32, 96, 5, 9, 11)])
def test_conv3d_cat_ops_match_concatenated_tensor(N, Ca, Cb, K, D, H, W):
    """lnn_conv3d_{fwd,dgrad,wgrad}_cat on two separate tensors == the plain ops on their channel concatenation,
    bit for bit (same kernels, same arithmetic order; only the addressing differs)."""
    C = Ca + Cb
    x = _rand((N, C, D, H, W), 1)
    w = _rand((K, C, 3, 3, 3), 2, 0.1).to(DEV)
    b = torch.randn(K, generator=torch.Generator().manual_seed(3)).to(DEV)
    dy = _rand((N, K, D, H, W), 4)
    xb, _ = to_cl_h(x)
    ld = max(Ca, Cb)
    xa = torch.zeros((N, D, H, W, ld), dtype=torch.float16, device=DEV); xa[..., :Ca] = xb[..., :Ca]
    xc = torch.zeros((N, D, H, W, ld), dtype=torch.float16, device=DEV); xc[..., :Cb] = xb[..., Ca:]
    dyb, _ = to_cl_h(dy)
    wf, wd = pack_conv_fwd(w), pack_conv_dgrad(w)
    for which in (-1, 5, 9, 10):
        assert nat.lib().lnn_debug_force_conv_kernel(which) == 0
        try:
            y1 = torch.empty((N, D, H, W, K), dtype=torch.float16, device=DEV); y2 = torch.empty_like(y1)
            nat.call("lnn_conv3d_fwd", xb, C, wf, b, y1, K, N, D, H, W, C, K, 1)
            nat.call("lnn_conv3d_fwd_cat", xa, xc, ld, Ca, wf, b, y2, K, N, D, H, W, C, K)
            assert torch.equal(y1, y2)
            for acc in (0, 1):
                base = to_cl_h(_rand((N, C, D, H, W), 5))[0]
                dx1 = base.clone()
                da = torch.zeros((N, D, H, W, ld), dtype=torch.float16, device=DEV); da[..., :Ca] = base[..., :Ca]
                dc = torch.zeros((N, D, H, W, ld), dtype=torch.float16, device=DEV); dc[..., :Cb] = base[..., Ca:]
                nat.call("lnn_conv3d_dgrad", dyb, K, wd, dx1, C, N, D, H, W, C, K, 1, acc)
                nat.call("lnn_conv3d_dgrad_cat", dyb, K, wd, da, dc, ld, Ca, N, D, H, W, C, K, acc)
                assert torch.equal(dx1[..., :Ca], da[..., :Ca]) and torch.equal(dx1[..., Ca:], dc[..., :Cb])
        finally:
            nat.lib().lnn_debug_force_conv_kernel(-1)
    p1 = torch.zeros(nat.query("lnn_wgrad_panel_elems", 27, K, C), device=DEV); p2 = torch.zeros_like(p1)
    nat.call("lnn_conv3d_wgrad", xb, C, dyb, K, p1, N, D, H, W, C, K, 1)
    nat.call("lnn_conv3d_wgrad_cat", xa, xc, ld, Ca, dyb, K, p2, N, D, H, W, C, K)
    assert float((p1 - p2).abs().max()) <= 1e-5 * float(p1.abs().max())      # fp32 atomics: order differs between launches


@pytest.mark.parametrize("C,K,D,H,W", [(256, 256, 8, 12, 8), (320, 320, 5, 6, 5), (640, 320, 5, 6, 5), (128, 256, 16, 8, 8)])
def test_conv_splitk_small_layers(C, K, D, H, W):
    """Split-K path of the small deep layers (lnn_conv3d_fwd_in_stats / lnn_conv3d_dgrad_ws with an fp32 workspace): same
    result as the plain launch up to the fp32 summation order (one fp16 ulp), scratch contents irrelevant, accumulate honoured,
    and the split result is bit-reproducible (slices are added in a fixed order)."""
    torch.manual_seed(C + K + D)
    N = 2
    x = (torch.randn(N, D, H, W, C, device=DEV) * 0.5).half()
    w = torch.randn(K, C, 3, 3, 3, device=DEV) * (2.0 / (27 * C)) ** 0.5
    b = torch.randn(K, device=DEV) * 0.1
    wp = torch.empty(nat.query("lnn_packed_weight_elems", 27, K, C), dtype=torch.float16, device=DEV)
    nat.call("lnn_pack_weights", w, wp, 27, K, C, C * 27, 27, 1)
    wd = torch.empty(nat.query("lnn_packed_weight_elems", 27, C, K), dtype=torch.float16, device=DEV)
    nat.call("lnn_pack_weights", w, wd, 27, C, K, 27, C * 27, 1)
    ws = torch.zeros(nat.query("lnn_instnorm_ws_doubles", N, K), dtype=torch.float64, device=DEV)
    sk = torch.full((8 * N * D * H * W * max((C + 31) // 32, (K + 31) // 32) * 32,), float('nan'), device=DEV)   # contents must not matter
    y0, y1 = torch.empty(N, D, H, W, K, dtype=torch.float16, device=DEV), torch.empty(N, D, H, W, K, dtype=torch.float16, device=DEV)
    m0, r0, m1, r1 = (torch.empty(N * K, device=DEV) for _ in range(4))
    nat.call("lnn_conv3d_fwd_in_stats", x, None, C, 0, wp, b, y0, N, D, H, W, C, K, 1, 1e-5, m0, r0, ws, None, 0)
    nat.call("lnn_conv3d_fwd_in_stats", x, None, C, 0, wp, b, y1, N, D, H, W, C, K, 1, 1e-5, m1, r1, ws, sk, sk.numel())
    torch.cuda.synchronize()
    scale = float(y0.float().abs().max())
    assert float((y0.float() - y1.float()).abs().max()) <= 2e-3 * scale
    assert torch.allclose(m0, m1, atol=1e-3 * scale) and torch.allclose(r0, r1, rtol=2e-3)
    y2 = torch.empty_like(y1)
    nat.call("lnn_conv3d_fwd_in_stats", x, None, C, 0, wp, b, y2, N, D, H, W, C, K, 1, 1e-5, m1, r1, ws, sk, sk.numel())
    assert torch.equal(y1, y2)
    # against the CPU reference of the op
    ref = torch.nn.functional.conv3d(x.float().cpu().permute(0, 4, 1, 2, 3), w.half().float().cpu(), b.cpu(), padding=1).permute(0, 2, 3, 4, 1)
    assert float((y1.float().cpu() - ref).abs().max()) <= 4e-3 * float(ref.abs().max())
    # data gradient with accumulation into an existing tensor
    dy = (torch.randn(N, D, H, W, K, device=DEV) * 0.5).half()
    base = (torch.randn(N, D, H, W, C, device=DEV) * 0.5).half()
    g0, g1 = base.clone(), base.clone()
    nat.call("lnn_conv3d_dgrad", dy, K, wd, g0, C, N, D, H, W, C, K, 1, 1)
    nat.call("lnn_conv3d_dgrad_ws", dy, K, wd, g1, C, N, D, H, W, C, K, 1, 1, sk, sk.numel())
    torch.cuda.synchronize()
    gs = float(g0.float().abs().max())
    assert float((g0.float() - g1.float()).abs().max()) <= 2e-3 * gs


@pytest.mark.parametrize("case", ["c1", "s1_small", "s1_panels", "s1_cat", "s2", "convT"])
def test_wgrad_deterministic_variants(case):
    """lnn_*_wgrad_det (per-writer panel copies + ordered reduction): equal to the atomics path up to the fp32 summation order,
    and BIT-identical between two runs on a scratch full of NaNs."""
    torch.manual_seed(7)
    N = 2
    scratch = torch.full((64 * 1024 * 1024,), float("nan"), device=DEV)
    if case == "convT":
        C, K, D, H, W = 64, 32, 6, 9, 10
        x = (torch.randn(N, D, H, W, C, device=DEV) * 0.5).half()
        dy = (torch.randn(N, 2 * D, 2 * H, 2 * W, K, device=DEV) * 0.5).half()
        elems = nat.query("lnn_wgrad_panel_elems", 8, C, K)
        plain = lambda pn: nat.call("lnn_convT3d_k2s2_wgrad", x, C, dy, K, pn, N, D, H, W, C, K)
        det = lambda pn: nat.call("lnn_convT3d_k2s2_wgrad_det", x, C, dy, K, pn, N, D, H, W, C, K, scratch, scratch.numel())
    else:
        C, K, D, H, W, stride = {"c1": (1, 32, 9, 13, 17, 1), "s1_small": (32, 32, 9, 13, 17, 1), "s1_panels": (128, 64, 8, 9, 10, 1),
                                 "s1_cat": (64, 32, 9, 13, 17, 1), "s2": (32, 64, 10, 14, 18, 2)}[case]
        Do, Ho, Wo = [(v - 1) // stride + 1 for v in (D, H, W)]
        x = (torch.randn(N, D, H, W, C, device=DEV) * 0.5).half() if C > 1 else (torch.randn(N, D, H, W, device=DEV) * 0.5).half()
        dy = (torch.randn(N, Do, Ho, Wo, K, device=DEV) * 0.5).half()
        elems = nat.query("lnn_wgrad_panel_elems", 27, K, C) if C > 1 else nat.query("lnn_wgrad_panel_elems", 1, K, 27)
        ld = C if C > 1 else 1
        if case == "s1_cat":
            xa, xb = x[..., :32].contiguous(), x[..., 32:].contiguous()
            plain = lambda pn: nat.call("lnn_conv3d_wgrad_cat", xa, xb, 32, 32, dy, K, pn, N, D, H, W, C, K)
            det = lambda pn: nat.call("lnn_conv3d_wgrad_cat_det", xa, xb, 32, 32, dy, K, pn, N, D, H, W, C, K, scratch, scratch.numel())
        else:
            plain = lambda pn: nat.call("lnn_conv3d_wgrad", x, ld, dy, K, pn, N, D, H, W, C, K, stride)
            det = lambda pn: nat.call("lnn_conv3d_wgrad_det", x, ld, dy, K, pn, N, D, H, W, C, K, stride, scratch, scratch.numel())
    p0, p1, p2 = (torch.full((elems,), 0.25, device=DEV) for _ in range(3))     # the panels accumulate: a non-zero start
    plain(p0); det(p1); det(p2)
    torch.cuda.synchronize()
    assert torch.isfinite(p1).all()
    assert torch.equal(p1, p2)
    assert float((p0 - p1).norm() / p0.norm()) < 1e-5
    # an insufficient scratch is reported, never silently ignored
    with pytest.raises(RuntimeError):
        if case == "convT":
            nat.call("lnn_convT3d_k2s2_wgrad_det", x, C, dy, K, p1, N, D, H, W, C, K, scratch, 1024)
        elif case == "s1_cat":
            nat.call("lnn_conv3d_wgrad_cat_det", xa, xb, 32, 32, dy, K, p1, N, D, H, W, C, K, scratch, 1024)
        else:
            nat.call("lnn_conv3d_wgrad_det", x, ld, dy, K, p1, N, D, H, W, C, K, stride, scratch, 1024)


# The deep stride-1 layers of BASELINE configs[1] at exactly the bench shapes (N = 2; 160x192x160 plan, levels 3-5):
# name, Ca (+ Cb for a concatenated input), K, D, H, W
DEEP_LAYERS = [("enc3.1 256->256", 256, 0, 256, 20, 24, 20), ("dec1.0 512->256 cat", 256, 256, 256, 20, 24, 20),
               ("enc4.1 320->320", 320, 0, 320, 10, 12, 10), ("dec0.0 640->320 cat", 320, 320, 320, 10, 12, 10),
               ("enc5.1 320->320", 320, 0, 320, 5, 6, 5)]


@pytest.mark.parametrize("name,Ca,Cb,K,D,H,W", DEEP_LAYERS, ids=[c[0].split()[0] for c in DEEP_LAYERS])
@pytest.mark.parametrize("which", [-1, 10])
def test_deep_layers_at_bench_shapes(name, Ca, Cb, K, D, H, W, which):
    """Forward (+ bias, split-K workspace as the engine passes it) and data gradient of the level 3-5 stride-1 convolutions at the
    BASELINE shapes against F.conv3d on the CPU in fp32 (2e-3 / 3e-3: fp16 storage, fp32 accumulation in a different order),
    with the automatic kernel selection and with the macro-tile kernel (igemm_conv_mt.hip) pinned.  Concatenated inputs go through
    the *_cat entries (two tensors, never materialised), their data gradient through lnn_conv3d_dgrad_cat_ws.
    Layers: test/network_architecture/test_MultiHead_Module.py:346-415; forward order generic_ViT_UNet.py:261-286."""
    N, C = 2, Ca + Cb
    g = torch.Generator().manual_seed(C + K + D)
    x = torch.randn((N, C, D, H, W), generator=g) * 0.5
    w = torch.randn((K, C, 3, 3, 3), generator=g) * (2.0 / (27 * C)) ** 0.5
    b = torch.randn(K, generator=g) * 0.1
    dy = torch.randn((N, K, D, H, W), generator=g) * 0.5
    xq, wq, dyq = x.half().float(), w.half().float(), dy.half().float()
    ref = F.conv3d(xq, wq, b, padding=1)
    ref_dx = F.conv_transpose3d(dyq, wq, None, padding=1)
    xb, _ = to_cl_h(x)
    dyb, _ = to_cl_h(dy)
    wf, wd = pack_conv_fwd(w.to(DEV)), pack_conv_dgrad(w.to(DEV))
    sk = torch.full((8 * N * D * H * W * ((max(C, K) + 31) // 32) * 32,), float("nan"), device=DEV)      # contents must not matter
    ws = torch.zeros(nat.query("lnn_instnorm_ws_doubles", N, K), dtype=torch.float64, device=DEV)
    mean, rstd = torch.zeros(N * K, device=DEV), torch.zeros(N * K, device=DEV)
    y = torch.full((N, D, H, W, K), 7.0, dtype=torch.float16, device=DEV)
    assert nat.lib().lnn_debug_force_conv_kernel(which) == 0
    try:
        if Cb:
            xa, xc = xb[..., :Ca].contiguous(), xb[..., Ca:].contiguous()
            nat.call("lnn_conv3d_fwd_in_stats", xa, xc, max(Ca, Cb), Ca, wf, b.to(DEV), y, N, D, H, W, C, K, 1, 1e-5, mean, rstd, ws, sk, sk.numel())
            da = torch.full((N, D, H, W, Ca), 7.0, dtype=torch.float16, device=DEV); dc = torch.full((N, D, H, W, Cb), 7.0, dtype=torch.float16, device=DEV)
            nat.call("lnn_conv3d_dgrad_cat_ws", dyb, K, wd, da, dc, max(Ca, Cb), Ca, N, D, H, W, C, K, 0, sk, sk.numel())
            dx = torch.cat([da, dc], dim=-1)
        else:
            nat.call("lnn_conv3d_fwd_in_stats", xb, None, C, 0, wf, b.to(DEV), y, N, D, H, W, C, K, 1, 1e-5, mean, rstd, ws, sk, sk.numel())
            dx = torch.full((N, D, H, W, C), 7.0, dtype=torch.float16, device=DEV)
            nat.call("lnn_conv3d_dgrad_ws", dyb, K, wd, dx, C, N, D, H, W, C, K, 1, 0, sk, sk.numel())
        torch.cuda.synchronize()
    finally:
        nat.lib().lnn_debug_force_conv_kernel(-1)
    assert rel_err(from_cl_h(y, K), ref) < 2e-3
    assert rel_err(from_cl_h(dx, C), ref_dx) < 3e-3
    # the statistics the call returns are those of the stored tensor
    yf = y.float().reshape(N, D * H * W, K)
    assert float((mean - yf.mean(1).reshape(-1)).abs().max()) <= 1e-4 * float(yf.abs().max())


@pytest.mark.parametrize("C,K,D,H,W,acc", [(32, 96, 3, 9, 10, 0), (64, 64, 5, 12, 10, 1), (16, 40, 3, 24, 20, 1), (48, 32, 7, 6, 5, 0),
                                            (32, 64, 4, 8, 40, 0), (32, 64, 3, 10, 64, 0), (48, 40, 4, 7, 45, 1), (16, 32, 2, 20, 128, 0)])
def test_macro_tile_kernel_split_k_and_ragged_bands(C, K, D, H, W, acc):
    """igemm_conv_mt.hip pinned on small volumes: odd plane counts (the last band has one live plane), a last row band shorter than
    TY, output channels that are not a multiple of 64 (the second row block of a channel block is dead), 40-wide rows (the widest
    band the 32 KB halo image holds in one piece), wider planes cut into column bands (64 -> 2 x 32, 128 -> 4 x 32, a ragged 45 -> 3 x 15), the K split over chunks with a NaN-filled workspace (fixed-order finalize: bit-reproducible),
    accumulate into an existing gradient."""
    N = 2
    g = torch.Generator().manual_seed(C * K + W)
    x = torch.randn((N, C, D, H, W), generator=g) * 0.5
    w = torch.randn((K, C, 3, 3, 3), generator=g) * 0.1
    b = torch.randn(K, generator=g)
    dy = torch.randn((N, K, D, H, W), generator=g) * 0.5
    base = torch.randn((N, C, D, H, W), generator=g)
    ref = F.conv3d(x.half().float(), w.half().float(), b, padding=1)
    ref_dx = F.conv_transpose3d(dy.half().float(), w.half().float(), None, padding=1) + (base.half().float() if acc else 0)
    xb, _ = to_cl_h(x)
    dyb, _ = to_cl_h(dy)
    wf, wd = pack_conv_fwd(w.to(DEV)), pack_conv_dgrad(w.to(DEV))
    sk = torch.full((16 * N * D * H * W * ((max(C, K) + 31) // 32) * 32,), float("nan"), device=DEV)
    ws = torch.zeros(nat.query("lnn_instnorm_ws_doubles", N, K), dtype=torch.float64, device=DEV)
    mean, rstd = torch.zeros(N * K, device=DEV), torch.zeros(N * K, device=DEV)
    ys, dxs = [], []
    assert nat.lib().lnn_debug_force_conv_kernel(10) == 0
    try:
        for wsbuf in (None, sk, sk):
            y = torch.full((N, D, H, W, K), 7.0, dtype=torch.float16, device=DEV)
            nat.call("lnn_conv3d_fwd_in_stats", xb, None, C, 0, wf, b.to(DEV), y, N, D, H, W, C, K, 1, 1e-5, mean, rstd, ws,
                     wsbuf, 0 if wsbuf is None else wsbuf.numel())
            dx, _ = to_cl_h(base, ld=C + 8)
            nat.call("lnn_conv3d_dgrad_ws", dyb, K, wd, dx, C + 8, N, D, H, W, C, K, 1, acc, wsbuf, 0 if wsbuf is None else wsbuf.numel())
            ys.append(y); dxs.append(dx)
        torch.cuda.synchronize()
    finally:
        nat.lib().lnn_debug_force_conv_kernel(-1)
    for y, dx in zip(ys, dxs):
        assert rel_err(from_cl_h(y, K), ref) < 2e-3
        assert rel_err(from_cl_h(dx, C), ref_dx) < 3e-3
    assert torch.equal(ys[1], ys[2]) and torch.equal(dxs[1], dxs[2])        # split-K: slices added in a fixed order
    if not acc:
        assert torch.all(dxs[0][..., C:] == 0)                                   # channel padding of the buffer untouched

