"""One-launch normalisation of the small volumes (-m gpu): lnn_conv3d_fwd_in_lrelu / lnn_conv3d_dgrad_in_bwd (csrc/norm_act.hip
in_small_*_kernel behind csrc/igemm_conv.hip) with the library's DEFAULT kernel selection -- at the bench shapes of the two lowest
levels the convolutions in front are the macro-tile kernel and the flattened-voxel kernels, with and without their split-K workspace.  Reference ops: nn.Conv3d -> nn.InstanceNorm3d(eps 1e-5, affine) ->
nn.LeakyReLU(1e-2) (ConvDropoutNormNonlin, test/network_architecture/test_MultiHead_Module.py:394-415) and their autograd, CPU fp32."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from tests.gpu_utils import DEV, from_cl_h, pack_conv_dgrad, pack_conv_fwd, q16, rel_err, to_cl_h  # noqa: E402
from lifelong_nnunet_amd import native as nat  # noqa: E402


SMALL_BLOCKS = [(320, 320, 10, 12, 10, 1, 0), (640, 320, 10, 12, 10, 1, 1), (320, 320, 5, 6, 5, 1, 0), (256, 320, 20, 24, 20, 2, 0),
                (320, 320, 10, 12, 10, 2, 0), (32, 48, 3, 9, 10, 1, 0), (16, 24, 12, 13, 13, 1, 0), (64, 32, 16, 16, 8, 1, 0)]


@pytest.mark.parametrize("ws_on", [0, 1])
@pytest.mark.parametrize("C,K,D,H,W,s,cat", SMALL_BLOCKS)
def test_small_volume_block_forward_in_one_call(C, K, D, H, W, s, cat, ws_on):
    """lnn_conv3d_fwd_in_lrelu (the lowest levels at bench shapes N = 2 -- 320 -> 320 and 640 -> 320 cat @ 10x12x10, 320 -> 320 @ 5x6x5,
    the strided 256 -> 320 and 320 -> 320 -- ragged toy shapes, and a 2048-voxel block at the limit) == lnn_conv3d_fwd_in_stats +
    lnn_instnorm_lrelu_fwd: y bit for bit, mean / rstd to fp32 summation order, z to one fp16 ulp; and z against conv3d -> instance_norm -> leaky_relu on the CPU in fp32."""
    N = 2
    g = torch.Generator().manual_seed(C + K + D)
    x = q16(torch.randn(N, C, D, H, W, generator=g) * 0.7)
    w = q16(torch.randn(K, C, 3, 3, 3, generator=g) * (2.0 / (27 * C)) ** 0.5)
    b = torch.randn(K, generator=g) * 0.1
    gamma = 1 + 0.3 * torch.randn(K, generator=g); gamma[::7] *= -1.0
    beta = 0.2 * torch.randn(K, generator=g)
    xb, _ = to_cl_h(x)
    wp = pack_conv_fwd(w.to(DEV))
    Do, Ho, Wo = (D - 1) // s + 1, (H - 1) // s + 1, (W - 1) // s + 1
    V = Do * Ho * Wo
    assert V <= nat.query("lnn_instnorm_small_volume")
    ws = torch.zeros(nat.query("lnn_instnorm_ws_doubles", N, K), dtype=torch.float64, device=DEV)
    sk = torch.full((64 * N * V * ((K + 31) // 32) * 32,), float('nan'), device=DEV) if ws_on else None
    skn = sk.numel() if ws_on else 0
    bd, ga, be = b.to(DEV), gamma.to(DEV), beta.to(DEV)
    xa = xc = None
    if cat:
        xa, xc = xb[..., :C // 2].contiguous(), xb[..., C // 2:].contiguous()
    outs = []
    for one_call in (False, True):
        y = torch.full((N, Do, Ho, Wo, K), 3.0, dtype=torch.float16, device=DEV)
        z = torch.full((N, Do, Ho, Wo, K + 8), 5.0, dtype=torch.float16, device=DEV)
        m, r = torch.zeros(N * K, device=DEV), torch.zeros(N * K, device=DEV)
        a0, a1, ld, ca = (xa, xc, C // 2, C // 2) if cat else (xb, None, C, 0)
        if one_call:
            nat.call("lnn_conv3d_fwd_in_lrelu", a0, a1, ld, ca, wp, bd, y, N, D, H, W, C, K, s, 1e-5, m, r, ga, be, 0.01, z, K + 8, ws, sk, skn)
        else:
            nat.call("lnn_conv3d_fwd_in_stats", a0, a1, ld, ca, wp, bd, y, N, D, H, W, C, K, s, 1e-5, m, r, ws, sk, skn)
            nat.call("lnn_instnorm_lrelu_fwd", y, z, K + 8, N, V, K, m, r, ga, be, 0.01)
        outs.append((y, z, m, r))
    (y0, z0, m0, r0), (y1, z1, m1, r1) = outs
    assert torch.equal(y0, y1)
    assert float((m0 - m1).abs().max()) <= 2e-6 * float(m0.abs().max()) + 1e-7 and float((r0 - r1).abs().max()) <= 2e-6 * float(r0.abs().max())
    assert bool((z1[..., K:] == 5.0).all())
    zs = float(z0.float().abs().max())
    assert float((z0[..., :K].float() - z1[..., :K].float()).abs().max()) <= 1e-3 * zs
    ref = F.leaky_relu(F.instance_norm(F.conv3d(x, w, b, stride=s, padding=1), weight=gamma, bias=beta, eps=1e-5), 0.01)
    assert rel_err(from_cl_h(z1, K), ref) < 4e-3


@pytest.mark.parametrize("ws_on", [0, 1])
@pytest.mark.parametrize("C,D,H,W", [(320, 10, 12, 10), (320, 5, 6, 5), (48, 3, 9, 10), (24, 12, 13, 13), (32, 16, 16, 8)])
def test_small_volume_block_backward_in_one_call(C, D, H, W, ws_on):
    """lnn_conv3d_dgrad_in_bwd: the data gradient of a stride-1 convolution + the WHOLE InstanceNorm / LeakyReLU backward of the block
    that produced its input (reduce, sums and apply as one launch).  Against
    autograd on the CPU in fp32: dL/du in place over u, the affine gradients (added to what is there, unscaled by grad_unscale), the
    (sample, channel) sums; and against the multi-launch passes on the same inputs (lnn_instnorm_lrelu_bwd takes the one-launch
    kernel on these volumes too: both routes, with and without the fp32 workspace, agree to an fp16 ulp)."""
    N, K = 2, C
    g = torch.Generator().manual_seed(C + D)
    u = q16(torch.randn(N, C, D, H, W, generator=g) * 1.5 + 0.3).requires_grad_(True)
    gamma = (1 + 0.3 * torch.randn(C, generator=g)); gamma[::5] *= -1.0
    gamma = gamma.requires_grad_(True)
    beta = (0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    w = q16(torch.randn(K, C, 3, 3, 3, generator=g) * (2.0 / (27 * C)) ** 0.5)
    z = F.leaky_relu(F.instance_norm(u, weight=gamma, bias=beta, eps=1e-5), 0.01)
    y2 = F.conv3d(z, w, None, padding=1)
    dy = q16(torch.randn(y2.shape, generator=g))
    y2.backward(dy)
    V = D * H * W
    dyb, _ = to_cl_h(dy)
    wp = pack_conv_dgrad(w.to(DEV))
    mean = torch.empty(N * C, device=DEV); rstd = torch.empty(N * C, device=DEV)
    nws = nat.query("lnn_instnorm_ws_doubles", N, C)
    ub0, _ = to_cl_h(u.detach())
    nat.call("lnn_instnorm_stats", ub0, N, V, C, 1e-5, mean, rstd, torch.zeros(nws, dtype=torch.float64, device=DEV))
    ga, be = gamma.detach().to(DEV), beta.detach().to(DEV)
    sk = torch.full((64 * N * V * ((C + 31) // 32) * 32,), float('nan'), device=DEV) if ws_on else None
    skn = sk.numel() if ws_on else 0
    res = []
    for one_call in (False, True):
        ub = ub0.clone()
        dx = torch.full((N, D, H, W, C), 7.0, dtype=torch.float16, device=DEV)
        ws = torch.zeros(nws, dtype=torch.float64, device=DEV)
        dg = torch.full((C,), 0.25, device=DEV); db = torch.full((C,), -0.5, device=DEV)
        if one_call:
            nat.call("lnn_conv3d_dgrad_in_bwd", dyb, K, wp, dx, C, N, D, H, W, C, K, ub, mean, rstd, ga, be, 0.01, dg, db, 0.5, ws, sk, skn)
        else:
            nat.call("lnn_conv3d_dgrad_ws", dyb, K, wp, dx, C, N, D, H, W, C, K, 1, 0, sk, skn)
            nat.call("lnn_instnorm_lrelu_bwd", ub, dx, C, N, V, C, mean, rstd, ga, be, 0.01, dg, db, None, 0.5, ws)
        res.append((ub, ws[:N * C * 3].view(N * C, 3)[:, :2].clone(), dg, db, dx))
    (u0, s0, dg0, db0, dx0), (u1, s1, dg1, db1, dx1) = res
    assert torch.equal(dx0, dx1)                                   # dL/dz of the data gradient, both routes
    us = float(u0.float().abs().max())
    assert float((u0.float() - u1.float()).abs().max()) <= 1e-3 * us
    scale = s0.abs().max(0).values
    assert float(((s0 - s1).abs() / scale).max()) < 2e-5
    assert rel_err(dg1.cpu(), dg0.cpu()) < 2e-5 and rel_err(db1.cpu(), db0.cpu()) < 2e-5
    assert rel_err(dg1.cpu() - 0.25, 0.5 * gamma.grad) < 3e-3 and rel_err(db1.cpu() + 0.5, 0.5 * beta.grad) < 3e-3
    assert rel_err(from_cl_h(u1, C), u.grad) < 4e-3


def test_small_volume_entry_refuses_large_volumes():
    x = torch.zeros(1, 16, 16, 16, 32, dtype=torch.float16, device=DEV)
    f = torch.zeros(64, device=DEV)
    ws = torch.zeros(nat.query("lnn_instnorm_ws_doubles", 1, 32), dtype=torch.float64, device=DEV)
    with pytest.raises(RuntimeError, match="voxels per sample"):
        nat.call("lnn_conv3d_dgrad_in_bwd", x, 32, x, x, 32, 1, 16, 16, 16, 32, 32, x, f, f, f, f, 0.01, f, f, 1.0, ws, None, 0)


@pytest.mark.parametrize("N,C,D,H,W", [(3, 8, 2, 2, 2), (1, 40, 1, 1, 7), (3, 72, 4, 16, 32), (2, 320, 5, 6, 5)])
def test_small_volume_norm_backward_odd_shapes(N, C, D, H, W):
    """lnn_instnorm_lrelu_bwd on volumes up to lnn_instnorm_small_volume() (one launch: csrc/norm_act.hip in_small_bwd_kernel) with
    channel counts that are not multiples of 32 (idle octet lanes), 8 voxels, exactly 2048 voxels, three samples (dgamma / dbeta are an
    ordered sum over the samples): against autograd through leaky_relu(instance_norm(u)) on the CPU in fp32."""
    g = torch.Generator().manual_seed(N * C + W)
    u = q16(torch.randn(N, C, D, H, W, generator=g) * 1.5 + 0.3).requires_grad_(True)
    gamma = (1 + 0.3 * torch.randn(C, generator=g)); gamma[::3] *= -1.0
    gamma = gamma.requires_grad_(True)
    beta = (0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    z = F.leaky_relu(F.instance_norm(u, weight=gamma, bias=beta, eps=1e-5), 0.01)
    dz = q16(torch.randn(z.shape, generator=g))
    z.backward(dz)
    V = D * H * W
    assert V <= nat.query("lnn_instnorm_small_volume")
    ub, _ = to_cl_h(u.detach())
    dzb, _ = to_cl_h(dz, ld=C + 8)
    mean = torch.empty(N * C, device=DEV); rstd = torch.empty(N * C, device=DEV)
    ws = torch.zeros(nat.query("lnn_instnorm_ws_doubles", N, C), dtype=torch.float64, device=DEV)
    nat.call("lnn_instnorm_stats", ub, N, V, C, 1e-5, mean, rstd, ws)
    dg = torch.full((C,), 0.5, device=DEV); db = torch.full((C,), -1.0, device=DEV)
    nat.call("lnn_instnorm_lrelu_bwd", ub, dzb, C + 8, N, V, C, mean, rstd, gamma.detach().to(DEV), beta.detach().to(DEV), 0.01, dg, db,
             None, 2.0, ws)
    tol = 2e-2 if V < 16 else 4e-3          # 8 voxels: the statistics themselves carry the fp16 rounding of u
    assert rel_err(from_cl_h(ub, C), u.grad) < tol
    assert rel_err(dg.cpu() - 0.5, 2.0 * gamma.grad) < tol and rel_err(db.cpu() + 1.0, 2.0 * beta.grad) < tol
