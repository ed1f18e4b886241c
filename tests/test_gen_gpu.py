"""-m gpu: the generic-geometry kernels (csrc/igemm_gen.hip) -- kernel extent 1 or 3 and stride 1 or 2 PER AXIS, transposed
convolutions with kernel == stride per axis -- against torch's CPU fp32 ops (what the reference reaches for a plan with
anisotropic ``conv_kernel_sizes`` / ``pool_op_kernel_sizes``, nnUNetTrainerMultiHead.py:348-369), through the C-ABI:
the ``lnn_*_g`` entries directly, and the isotropic entries with the generic kernels forced (``lnn_debug_set_gen_mode(1)``; by
default they take the volumes of at most 4096 output voxels).  Inputs are pre-rounded to fp16; tolerances as in
tests/test_kernels_gpu.py (fp16 output rounding + fp32 accumulation order)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from tests.gpu_utils import DEV, View, from_cl_h, pack, q16, rel_err, to_cl_h      # noqa: E402
from lifelong_nnunet_amd import native as nat                                      # noqa: E402


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return q16(torch.randn(shape, generator=g) * scale)


def _ws(n=1 << 22):
    return torch.full((n,), float("nan"), device=DEV)          # the kernels must not read what they did not write


GEN_CONV_CASES = [
    # N, C, K, D, H, W, kernel, stride
    (2, 32, 32, 6, 10, 12, (3, 3, 3), (1, 1, 1)),
    (1, 24, 40, 5, 9, 11, (1, 3, 3), (1, 2, 2)),        # ragged channel chunk (24 = 16 + 8), 40 output channels, Prostate-like geometry
    (2, 8, 8, 4, 6, 6, (3, 3, 1), (2, 2, 1)),
    (1, 64, 96, 3, 7, 5, (3, 3, 3), (2, 2, 2)),         # odd extents at stride 2
    (1, 16, 16, 4, 4, 4, (1, 1, 1), (1, 1, 1)),
    (2, 320, 320, 5, 6, 5, (3, 3, 3), (1, 1, 1)),       # the bottleneck of the 160x192x160 plan
    (2, 32, 64, 7, 12, 10, (3, 3, 3), (1, 2, 2)),
    (1, 48, 32, 9, 5, 4, (3, 1, 3), (2, 1, 2)),
]


@pytest.mark.parametrize("N,C,K,D,H,W,k,st", GEN_CONV_CASES)
@pytest.mark.parametrize("split", [False, True])
def test_gen_conv_fwd(N, C, K, D, H, W, k, st, split):
    nt = k[0] * k[1] * k[2]
    x = _rand((N, C, D, H, W), 1)
    w = _rand((K, C) + k, 2, 0.1)
    b = torch.randn(K, generator=torch.Generator().manual_seed(3))
    ref = F.conv3d(x, w, b, stride=st, padding=tuple(a // 2 for a in k))
    xb, _ = to_cl_h(x, ld=C + 8, offset=8)
    Do, Ho, Wo = ref.shape[2:]
    yb = torch.full((N, Do, Ho, Wo, K + 16), 7.0, dtype=torch.float16, device=DEV)
    wp = pack(w.to(DEV), nt, K, C, C * nt, nt, 1)
    ws = _ws() if split else None
    nat.call("lnn_conv3d_fwd_g", View(xb, 8), C + 8, wp, b.to(DEV), View(yb, 16), K + 16, N, D, H, W, C, K, *k, *st,
             ws, 0 if ws is None else ws.numel())
    assert rel_err(from_cl_h(yb, K, 16), ref) < 2e-3
    assert torch.all(yb[..., :16] == 7.0)


@pytest.mark.parametrize("N,C,K,D,H,W,k,st", GEN_CONV_CASES)
@pytest.mark.parametrize("acc,split", [(0, False), (1, False), (0, True), (1, True)])
def test_gen_conv_dgrad(N, C, K, D, H, W, k, st, acc, split):
    nt = k[0] * k[1] * k[2]
    x = _rand((N, C, D, H, W), 1).requires_grad_(True)
    w = _rand((K, C) + k, 2, 0.1)
    y = F.conv3d(x, w, None, stride=st, padding=tuple(a // 2 for a in k))
    dy = _rand(y.shape, 4)
    y.backward(dy)
    dyb, _ = to_cl_h(dy)
    base = _rand((N, C, D, H, W), 5)
    dxb, _ = to_cl_h(base, ld=C + 8)
    wp = pack(w.to(DEV), nt, C, K, nt, C * nt, 1)
    ws = _ws() if split else None
    nat.call("lnn_conv3d_dgrad_g", dyb, K, wp, dxb, C + 8, N, D, H, W, C, K, *k, *st, acc, ws, 0 if ws is None else ws.numel())
    exp = x.grad + base if acc else x.grad
    assert rel_err(from_cl_h(dxb, C), exp) < 3e-3


@pytest.mark.parametrize("N,C,K,D,H,W,k,st", GEN_CONV_CASES)
@pytest.mark.parametrize("det", [False, True])
def test_gen_conv_wgrad(N, C, K, D, H, W, k, st, det):
    nt = k[0] * k[1] * k[2]
    x = _rand((N, C, D, H, W), 1)
    w = _rand((K, C) + k, 2, 0.1).requires_grad_(True)
    y = F.conv3d(x, w, None, stride=st, padding=tuple(a // 2 for a in k))
    dy = _rand(y.shape, 4)
    y.backward(dy)
    xb, _ = to_cl_h(x, ld=C + 8)
    dyb, _ = to_cl_h(dy)
    panel = torch.zeros(nat.query("lnn_wgrad_panel_elems", nt, K, C), device=DEV)
    parts = _ws(1 << 24) if det else None
    runs = []
    for _ in range(2 if det else 1):
        panel.zero_()
        nat.call("lnn_conv3d_wgrad_g", xb, C + 8, dyb, K, panel, N, D, H, W, C, K, *k, *st, parts, 0 if parts is None else parts.numel())
        runs.append(panel.clone())
    dw = torch.full((K, C) + k, 1.0, device=DEV)
    nat.call("lnn_unpack_wgrad", panel, dw, nt, K, C, C * nt, nt, 1, 0.5, 1)
    assert rel_err(dw.cpu(), 1.0 + 0.5 * w.grad) < 1e-3
    if det:
        assert torch.equal(runs[0], runs[1])                    # ordered reduction: bit-reproducible


GEN_CONVT_CASES = [(2, 64, 32, 4, 6, 4, (2, 2, 2)), (1, 16, 8, 3, 5, 6, (1, 2, 2)), (1, 320, 320, 5, 6, 5, (2, 2, 2)),
                   (1, 32, 24, 4, 3, 5, (2, 2, 1)), (2, 320, 256, 10, 12, 10, (2, 2, 2))]


@pytest.mark.parametrize("N,C,K,D,H,W,st", GEN_CONVT_CASES)
@pytest.mark.parametrize("split", [False, True])
def test_gen_convT(N, C, K, D, H, W, st, split):
    nt = st[0] * st[1] * st[2]
    x = _rand((N, C, D, H, W), 1).requires_grad_(True)
    w = _rand((C, K) + st, 2, 0.1).requires_grad_(True)
    ref = F.conv_transpose3d(x, w, None, stride=st)
    dy = _rand(ref.shape, 4)
    ref.backward(dy)
    xb, _ = to_cl_h(x.detach())
    wd = w.detach().to(DEV)
    ws = _ws() if split else None
    wsn = 0 if ws is None else ws.numel()
    yb = torch.zeros((N,) + tuple(ref.shape[2:]) + (2 * K,), dtype=torch.float16, device=DEV)
    nat.call("lnn_convT3d_fwd_g", xb, C, pack(wd, nt, K, C, nt, K * nt, 1), yb, 2 * K, N, D, H, W, C, K, *st, ws, wsn)
    assert rel_err(from_cl_h(yb, K), ref.detach()) < 2e-3
    assert torch.all(yb[..., K:] == 0)
    dyb, _ = to_cl_h(dy)
    for acc in (0, 1):
        base = _rand((N, C, D, H, W), 6)
        dxb, _ = to_cl_h(base)
        nat.call("lnn_convT3d_dgrad_g", dyb, K, pack(wd, nt, C, K, K * nt, nt, 1), dxb, C, N, D, H, W, C, K, *st, acc, ws, wsn)
        assert rel_err(from_cl_h(dxb, C), x.grad + base if acc else x.grad) < 3e-3
    panel = torch.zeros(nat.query("lnn_wgrad_panel_elems", nt, C, K), device=DEV)
    nat.call("lnn_convT3d_wgrad_g", xb, C, dyb, K, panel, N, D, H, W, C, K, *st, None, 0)
    dw = torch.zeros((C, K) + st, device=DEV)
    nat.call("lnn_unpack_wgrad", panel, dw, nt, C, K, K * nt, nt, 1, 1.0, 0)
    assert rel_err(dw.cpu(), w.grad) < 1e-3


@pytest.fixture
def gen_forced():
    assert nat.lib().lnn_debug_set_gen_mode(1) == 0
    yield
    nat.lib().lnn_debug_set_gen_mode(-1)


@pytest.mark.parametrize("N,C,K,D,H,W,s", [(2, 32, 32, 8, 16, 8, 1), (1, 48, 96, 4, 8, 8, 1), (2, 64, 64, 9, 17, 5, 2), (1, 16, 32, 7, 9, 11, 2)])
def test_isotropic_entries_on_the_generic_kernels(gen_forced, N, C, K, D, H, W, s):
    """lnn_conv3d_fwd_in_stats / _dgrad_ws / _wgrad (+ _det) as the engine calls them, generic kernels forced."""
    x = _rand((N, C, D, H, W), 1).requires_grad_(True)
    w = _rand((K, C, 3, 3, 3), 2, 0.1).requires_grad_(True)
    b = torch.randn(K, generator=torch.Generator().manual_seed(3))
    ref = F.conv3d(x, w, b, stride=s, padding=1)
    dy = _rand(ref.shape, 4)
    ref.backward(dy)
    xb, _ = to_cl_h(x.detach())
    Do, Ho, Wo = ref.shape[2:]
    V = Do * Ho * Wo
    yb = torch.zeros((N, Do, Ho, Wo, K), dtype=torch.float16, device=DEV)
    mean, rstd = torch.empty(N * K, device=DEV), torch.empty(N * K, device=DEV)
    wsd = torch.zeros(nat.query("lnn_instnorm_ws_doubles", N, K), dtype=torch.float64, device=DEV)
    ws = _ws()
    wd = w.detach().to(DEV)
    nat.call("lnn_conv3d_fwd_in_stats", xb, None, C, 0, pack(wd, 27, K, C, C * 27, 27, 1), b.to(DEV), yb, N, D, H, W, C, K, s, 1e-5,
             mean, rstd, wsd, ws, ws.numel())
    got = from_cl_h(yb, K)
    assert rel_err(got, ref.detach()) < 2e-3
    assert rel_err(mean.cpu().view(N, K), got.mean((2, 3, 4))) < 1e-4          # statistics of the fp16 values it stored
    assert rel_err(rstd.cpu().view(N, K), 1 / torch.sqrt(got.var((2, 3, 4), unbiased=False) + 1e-5)) < 1e-4
    dyb, _ = to_cl_h(dy)
    dxb = torch.zeros((N, D, H, W, C), dtype=torch.float16, device=DEV)
    nat.call("lnn_conv3d_dgrad_ws", dyb, K, pack(wd, 27, C, K, 27, C * 27, 1), dxb, C, N, D, H, W, C, K, s, 0, ws, ws.numel())
    assert rel_err(from_cl_h(dxb, C), x.grad) < 3e-3
    panel = torch.zeros(nat.query("lnn_wgrad_panel_elems", 27, K, C), device=DEV)
    nat.call("lnn_conv3d_wgrad", xb, C, dyb, K, panel, N, D, H, W, C, K, s)
    dw = torch.zeros((K, C, 3, 3, 3), device=DEV)
    nat.call("lnn_unpack_wgrad", panel, dw, 27, K, C, C * 27, 27, 1, 1.0, 0)
    assert rel_err(dw.cpu(), w.grad) < 1e-3
    panel2 = torch.zeros_like(panel)
    parts = _ws(1 << 24)
    nat.call("lnn_conv3d_wgrad_det", xb, C, dyb, K, panel2, N, D, H, W, C, K, s, parts, parts.numel())
    assert rel_err(panel2.cpu(), panel.cpu()) < 1e-5


@pytest.mark.parametrize("N,C,K,D,H,W", [(2, 64, 32, 4, 8, 4), (1, 320, 320, 2, 3, 2), (2, 128, 64, 5, 3, 11)])
def test_isotropic_convT_entries_on_the_generic_kernels(gen_forced, N, C, K, D, H, W):
    x = _rand((N, C, D, H, W), 1).requires_grad_(True)
    w = _rand((C, K, 2, 2, 2), 2, 0.1).requires_grad_(True)
    ref = F.conv_transpose3d(x, w, None, stride=2)
    dy = _rand(ref.shape, 4)
    ref.backward(dy)
    xb, _ = to_cl_h(x.detach())
    wd = w.detach().to(DEV)
    ws = _ws()
    yb = torch.zeros((N, 2 * D, 2 * H, 2 * W, K), dtype=torch.float16, device=DEV)
    nat.call("lnn_convT3d_k2s2_fwd_ws", xb, C, pack(wd, 8, K, C, 8, K * 8, 1), yb, K, N, D, H, W, C, K, ws, ws.numel())
    assert rel_err(from_cl_h(yb, K), ref.detach()) < 2e-3
    dyb, _ = to_cl_h(dy)
    dxb = torch.zeros((N, D, H, W, C), dtype=torch.float16, device=DEV)
    nat.call("lnn_convT3d_k2s2_dgrad_ws", dyb, K, pack(wd, 8, C, K, K * 8, 8, 1), dxb, C, N, D, H, W, C, K, 0, ws, ws.numel())
    assert rel_err(from_cl_h(dxb, C), x.grad) < 3e-3
    panel = torch.zeros(nat.query("lnn_wgrad_panel_elems", 8, C, K), device=DEV)
    nat.call("lnn_convT3d_k2s2_wgrad", xb, C, dyb, K, panel, N, D, H, W, C, K)
    dw = torch.zeros((C, K, 2, 2, 2), device=DEV)
    nat.call("lnn_unpack_wgrad", panel, dw, 8, C, K, K * 8, 8, 1, 1.0, 0)
    assert rel_err(dw.cpu(), w.grad) < 1e-3
    parts = _ws(1 << 24)
    panel2 = torch.zeros_like(panel)
    nat.call("lnn_convT3d_k2s2_wgrad_det", xb, C, dyb, K, panel2, N, D, H, W, C, K, parts, parts.numel())
    assert rel_err(panel2.cpu(), panel.cpu()) < 1e-5


K133_CASES = [
    # N, C, K, D, H, W   (walk along H; footprint rows span D, columns W)
    (2, 32, 32, 5, 12, 18),         # ragged footprint rows (5 of 8) and columns (18 = 16 + 2), short walk
    (1, 64, 32, 9, 37, 9),          # the decoder's 64 -> 32: two chunk waves per footprint, odd walk length (z segments)
    (1, 64, 64, 4, 20, 24),         # 64 output channels per item
    (2, 32, 64, 7, 9, 33),          # the data-gradient shape of 64 -> 32
    (1, 128, 96, 6, 16, 10),        # eight chunk waves, three output blocks
    (1, 32, 32, 20, 40, 48),        # a slab of the Prostate-shaped top level
    (1, 32, 32, 1, 33, 40),         # a single plane (a 2-D image): one live footprint row
    (3, 64, 64, 3, 18, 5),          # fewer columns than a footprint is wide, three samples
]


@pytest.mark.parametrize("N,C,K,D,H,W", K133_CASES)
def test_k133_conv_on_the_z_streaming_kernel(N, C, K, D, H, W):
    """[1,3,3] stride-1 convolutions (the first stages of anisotropic plans, nnUNetTrainerMultiHead.py:348-369) on the z-streaming kernel
    with permuted axes (igemm_conv_v9.hip KY = 1: the walk runs along H, the three ky taps are its rolling accumulators), forced through
    lnn_debug_set_k133_v9(1): forward with bias and data gradient against torch's CPU fp32 ops, channel-padded tensors untouched
    outside their channels, and bit-equal to nothing else -- the flattened-voxel kernel (mode 0) must agree to fp16 rounding."""
    k, st = (1, 3, 3), (1, 1, 1)
    x = _rand((N, C, D, H, W), 1).requires_grad_(True)
    w = _rand((K, C) + k, 2, 0.1)
    b = torch.randn(K, generator=torch.Generator().manual_seed(3))
    ref = F.conv3d(x, w, b, stride=st, padding=(0, 1, 1))
    dy = _rand(ref.shape, 4)
    ref.backward(dy)
    xb, _ = to_cl_h(x.detach(), ld=C + 8, offset=8)
    dyb, _ = to_cl_h(dy)
    wpf = pack(w.to(DEV), 9, K, C, C * 9, 9, 1)
    wpd = pack(w.to(DEV), 9, C, K, 9, C * 9, 1)
    outs = {}
    try:
        for mode in (1, 0):
            assert nat.lib().lnn_debug_set_k133_v9(mode) == 0
            yb = torch.full((N, D, H, W, K + 16), 7.0, dtype=torch.float16, device=DEV)
            nat.call("lnn_conv3d_fwd_g", View(xb, 8), C + 8, wpf, b.to(DEV), View(yb, 16), K + 16, N, D, H, W, C, K, *k, *st, None, 0)
            assert nat.lib().lnn_debug_last_k133_on_v9() == mode
            dxb = torch.full((N, D, H, W, C + 8), 5.0, dtype=torch.float16, device=DEV)
            nat.call("lnn_conv3d_dgrad_g", dyb, K, wpd, dxb, C + 8, N, D, H, W, C, K, *k, *st, 0, None, 0)
            # (the data gradient gathers the K channels of dy: 96 is not one of the kernel's 32 / 64 / 128 -> the flattened-voxel kernel)
            assert nat.lib().lnn_debug_last_k133_on_v9() == (mode if K in (32, 64, 128) else 0)
            outs[mode] = (yb, dxb)
    finally:
        nat.lib().lnn_debug_set_k133_v9(-1)
    for mode in (1, 0):
        yb, dxb = outs[mode]
        assert rel_err(from_cl_h(yb, K, 16), ref) < 2e-3, mode
        assert torch.all(yb[..., :16] == 7.0)
        assert rel_err(from_cl_h(dxb, C), x.grad) < 3e-3, mode
        assert torch.all(dxb[..., C:] == 5.0)
    assert rel_err(from_cl_h(outs[1][0], K, 16), from_cl_h(outs[0][0], K, 16)) < 2e-3


@pytest.mark.parametrize("N,C,K,D,H,W", [(2, 32, 32, 5, 12, 18), (1, 64, 32, 9, 20, 9), (1, 16, 32, 4, 9, 24)])
def test_k133_weight_gradient_is_the_centre_slice_of_the_333_weight_gradient(N, C, K, D, H, W):
    """engine.ConvBlock.wgrad_333: dW of a [1,3,3] stride-1 convolution = taps 9..17 (kz = 1) of the 3x3x3 stride-1 weight gradient of
    the same x / dy (lnn_conv3d_wgrad, tap-major fp32 panel), unpacked as a 9-tap panel; against autograd on the CPU in fp32 and against
    lnn_conv3d_wgrad_g."""
    x = _rand((N, C, D, H, W), 1)
    w = _rand((K, C, 1, 3, 3), 2, 0.1).requires_grad_(True)
    y = F.conv3d(x, w, None, padding=(0, 1, 1))
    dy = _rand(y.shape, 4)
    y.backward(dy)
    xb, _ = to_cl_h(x)
    dyb, _ = to_cl_h(dy)
    p27 = torch.zeros(nat.query("lnn_wgrad_panel_elems", 27, K, C), device=DEV)
    nat.call("lnn_conv3d_wgrad", xb, C, dyb, K, p27, N, D, H, W, C, K, 1)
    mp, cp = -(-K // 32) * 32, -(-C // 32) * 32
    assert nat.query("lnn_wgrad_panel_elems", 9, K, C) == 9 * mp * cp
    dw = torch.zeros((K, C, 1, 3, 3), device=DEV)
    nat.call("lnn_unpack_wgrad", View(p27, 9 * mp * cp), dw, 9, K, C, C * 9, 9, 1, 1.0, 0)
    assert rel_err(dw.cpu(), w.grad) < 2e-3
    p9 = torch.zeros(9 * mp * cp, device=DEV)
    nat.call("lnn_conv3d_wgrad_g", xb, C, dyb, K, p9, N, D, H, W, C, K, 1, 3, 3, 1, 1, 1, None, 0)
    dg = torch.zeros_like(dw)
    nat.call("lnn_unpack_wgrad", p9, dg, 9, K, C, C * 9, 9, 1, 1.0, 0)
    assert rel_err(dw.cpu(), dg.cpu()) < 1e-3
