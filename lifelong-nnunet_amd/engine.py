"""Static launch plan for the 3-D Generic_UNet training step on one MI355X.

The network the reference trains (upstream ``Generic_UNet`` as configured at
nnViTUNetTrainer.py:101-125; forward restated at generic_ViT_UNet.py:222-230,261-286) is a fixed
sequence of ops for a fixed patch size, so the build runs it as a *plan*: a list of C-ABI launches over
pre-allocated channels-last fp16 buffers in HBM, enqueued on one HIP stream (graph-capturable).

HBM layout
  * parameters theta, gradients G, momentum: three flat fp32 arenas, laid out in FORWARD EXECUTION order
    (so gradient buckets complete back-to-front during backward, see parallel.py); every
    ``nn.Parameter`` of the host module is a view into theta, every ``.grad`` a view into G.
  * activations: NDHWC fp16.  Per ConvDropoutNormNonlin block: ``y`` (conv output, later overwritten
    in place by dL/dy) and ``z`` (normalised + LeakyReLU output).  Skip connections and transposed-conv
    outputs are written straight into the decoder's concat buffer (channel stride 2*C), which removes
    ``torch.cat`` (generic_ViT_UNet.py:263) from the path.
  * fp16 weight panels for the MFMA kernels (forward and dgrad orientation), re-packed from theta when
    the parameters change; fp32 weight-gradient panels (one arena, zeroed once per backward).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import os

import torch

from . import native as nat

LRELU_SLOPE = 1e-2
IN_EPS = 1e-5


class Act:
    """A channels-last fp16 activation living at a channel offset inside a (possibly wider) buffer."""

    def __init__(self, buf: torch.Tensor, off: int, C: int):
        self.buf, self.off, self.C = buf, off, C
        self.ld = buf.shape[-1]
        self.N = buf.shape[0]
        self.dims = tuple(buf.shape[1:4])
        self.V = self.dims[0] * self.dims[1] * self.dims[2]

    def data_ptr(self):
        return self.buf.data_ptr() + 2 * self.off

    def tensor(self):
        return self.buf[..., self.off:self.off + self.C]


@dataclass
class ParamSlot:
    name: str
    shape: tuple
    offset: int = 0

    @property
    def numel(self):
        n = 1
        for s in self.shape:
            n *= s
        return n


@dataclass
class ConvBlock:
    """conv(+bias) -> InstanceNorm(affine) -> LeakyReLU (ConvDropoutNormNonlin, dropout p=0 omitted).  ``kernel`` / ``strides``:
    per-axis extents from the plans (3x3x3 and stride 1 or 2 everywhere in the isotropic plans); ``stride`` is the common
    stride of an isotropic block (the specialised kernels), 0 otherwise (the generic-geometry kernels, csrc/igemm_gen.hip)."""
    prefix: str
    cin: int
    cout: int
    stride: int
    kernel: tuple = (3, 3, 3)
    strides: tuple = (1, 1, 1)
    cin_k: int = 0                 # channels of the input as the kernels see it (a multi-channel image is zero-padded to 16)
    first: bool = False            # the network's first convolution (no data gradient)
    x: Optional[Act] = None        # input activation (None -> the fp16 image, C == 1 path)
    gx: Optional[Act] = None       # gradient wrt input (None -> not needed)
    gx_accumulate: bool = False
    x2: Optional[Act] = None       # second half of a never-materialised channel concatenation (x = cat(x, x2))
    gx2: Optional[Act] = None
    y: Optional[torch.Tensor] = None
    z: Optional[Act] = None
    gz: Optional[Act] = None
    mean: Optional[torch.Tensor] = None
    rstd: Optional[torch.Tensor] = None
    w: ParamSlot = None
    b: ParamSlot = None
    gamma: ParamSlot = None
    beta: ParamSlot = None
    wp_fwd: int = 0
    wp_dgrad: int = 0
    panel: int = 0
    in_dims: tuple = ()
    x_block: object = None         # the ConvBlock whose output IS this block's only input (second block of a stage): its gz = gx
    wgrad_333: bool = False        # [1,3,3] stride-1 block whose weight gradient is taken as the kz = 1 slice of a 3x3x3 weight gradient

    @property
    def iso(self):
        return self.stride != 0

    @property
    def ntaps(self):
        return self.kernel[0] * self.kernel[1] * self.kernel[2]


@dataclass
class UpBlock:
    """ConvTranspose3d with kernel == stride == the pooling of its level, no bias (``tu``)."""
    prefix: str
    cin: int
    cout: int
    strides: tuple = (2, 2, 2)
    x: Act = None
    gx: Act = None
    y: Act = None
    gy: Act = None
    w: ParamSlot = None
    wp_fwd: int = 0
    wp_dgrad: int = 0
    panel: int = 0

    @property
    def iso(self):
        return tuple(self.strides) == (2, 2, 2)

    @property
    def ntaps(self):
        return self.strides[0] * self.strides[1] * self.strides[2]


@dataclass
class SegHead:
    prefix: str
    cin: int
    x: Act = None
    gx: Act = None
    gx_has_prior: bool = False     # a transposed-conv dgrad already wrote gx -> accumulate
    w: ParamSlot = None
    x_block: object = None         # the ConvBlock that produces x (its normalisation pass can compute the head as well)


@dataclass
class Plan:
    fwd: List[object] = field(default_factory=list)


def _cl(N, dims, C, device):
    return torch.zeros((N,) + tuple(dims) + (C,), dtype=torch.float16, device=device)


class ParamArena:
    """Flat fp32 parameter / gradient / momentum arenas, slots in forward execution order."""

    def __init__(self, slots: List[ParamSlot], device):
        off = 0
        for p in slots:
            p.offset = off
            off += (p.numel + 3) // 4 * 4          # 16-byte aligned slots
        self.slots = slots
        self.by_name = {p.name: p for p in slots}
        self.size = off
        self.theta = torch.zeros(off, device=device)
        self.grad = torch.zeros(off, device=device)
        self.momentum = torch.zeros(off, device=device)
        self.version = 0                            # bumped whenever theta changes (re-pack trigger)

    def view(self, name_or_slot, which="theta"):
        s = self.by_name[name_or_slot] if isinstance(name_or_slot, str) else name_or_slot
        return getattr(self, which)[s.offset:s.offset + s.numel].view(s.shape)


def unet_geometry(num_pool, patch_size=None, pool_op_kernel_sizes=None, conv_kernel_sizes=None):
    """(pools, kernels, dims) of a plan: ``pool_op_kernel_sizes`` (num_pool per-axis strides, default 2x2x2), ``conv_kernel_sizes``
    (num_pool + 1 per-axis kernel extents, default 3x3x3) as the reference hands them to Generic_UNet
    (nnUNetTrainerMultiHead.py:348-369 -> nnViTUNetTrainer.py:117-122), and the spatial extents of every level for ``patch_size``.
    Upstream Generic_UNet (nnunet @77bc485): encoder stage d uses kernel d and the stride of pooling d - 1 in its first block;
    the bottleneck kernel num_pool and pooling num_pool - 1; decoder stage u the transposed convolution of pooling -(u + 1) and
    -- as upstream indexes it -- conv kernel -(u + 1), i.e. the kernel of the encoder stage ONE LEVEL BELOW its resolution."""
    pools = [tuple(int(v) for v in q) for q in (pool_op_kernel_sizes if pool_op_kernel_sizes is not None else [(2, 2, 2)] * num_pool)]
    kernels = [tuple(int(v) for v in q) for q in (conv_kernel_sizes if conv_kernel_sizes is not None else [(3, 3, 3)] * (num_pool + 1))]
    assert len(pools) == num_pool and len(kernels) == num_pool + 1, "one pooling per level, one conv kernel per stage (+ bottleneck)"
    assert all(v in (1, 2) for q in pools for v in q), "pool_op_kernel_sizes entries must be 1 or 2"
    assert all(v in (1, 3) for q in kernels for v in q), "conv_kernel_sizes entries must be 1 or 3"
    dims = None
    if patch_size is not None:
        dims = [tuple(int(v) for v in patch_size)]
        for d in range(num_pool):
            assert all(dims[d][a] % pools[d][a] == 0 for a in range(3)), \
                f"patch size {tuple(patch_size)} must be divisible by the cumulative pooling strides"
            dims.append(tuple(dims[d][a] // pools[d][a] for a in range(3)))
    return pools, kernels, dims


def param_slots(in_channels, base_features, num_classes, num_pool, max_features=320, pool_op_kernel_sizes=None,
                conv_kernel_sizes=None) -> List[ParamSlot]:
    """Parameter tensors of Generic_UNet in FORWARD EXECUTION order (names as test_MultiHead_Module.py:282-431)."""
    feats = [min(base_features * 2 ** d, max_features) for d in range(num_pool + 1)]
    pools, kernels, _ = unet_geometry(num_pool, None, pool_op_kernel_sizes, conv_kernel_sizes)
    out = []

    def block(prefix, cin, cout, k):
        out.extend([ParamSlot(prefix + ".conv.weight", (cout, cin) + tuple(k)), ParamSlot(prefix + ".conv.bias", (cout,)),
                    ParamSlot(prefix + ".instnorm.weight", (cout,)), ParamSlot(prefix + ".instnorm.bias", (cout,))])

    cin = in_channels
    for d in range(num_pool):
        block(f"conv_blocks_context.{d}.blocks.0", cin, feats[d], kernels[d])
        block(f"conv_blocks_context.{d}.blocks.1", feats[d], feats[d], kernels[d])
        cin = feats[d]
    block(f"conv_blocks_context.{num_pool}.0.blocks.0", cin, feats[num_pool], kernels[num_pool])
    block(f"conv_blocks_context.{num_pool}.1.blocks.0", feats[num_pool], feats[num_pool], kernels[num_pool])
    cdown = feats[num_pool]
    for u in range(num_pool):
        cs = feats[num_pool - 1 - u]
        out.append(ParamSlot(f"tu.{u}.weight", (cdown, cs) + pools[-(u + 1)]))
        block(f"conv_blocks_localization.{u}.0.blocks.0", 2 * cs, cs, kernels[-(u + 1)])
        block(f"conv_blocks_localization.{u}.1.blocks.0", cs, cs, kernels[-(u + 1)])
        out.append(ParamSlot(f"seg_outputs.{u}.weight", (num_classes, cs, 1, 1, 1)))
        cdown = cs
    return out


class UNetEngine:
    def __init__(self, arena: ParamArena, in_channels, base_features, num_classes, num_pool, patch_size, batch_size,
                 device="cuda", max_features=320, conv_per_stage=2, pool_op_kernel_sizes=None, conv_kernel_sizes=None):
        assert conv_per_stage == 2, "nnUNetTrainerV2 uses conv_per_stage=2 (nnViTUNetTrainer.py:119)"
        assert base_features % 8 == 0, "channel counts must be multiples of 8 (16-byte vectors)"
        self.arena = arena
        self._pviews = {}
        pools, kernels, dims = unet_geometry(num_pool, patch_size, pool_op_kernel_sizes, conv_kernel_sizes)
        self.pools, self.kernels = pools, kernels
        self.in_channels, self.base, self.K, self.num_pool = in_channels, base_features, num_classes, num_pool
        self.patch, self.N, self.device = tuple(patch_size), batch_size, torch.device(device)
        self.max_features = max_features
        dev, N = self.device, batch_size

        feats = [min(base_features * 2 ** d, max_features) for d in range(num_pool + 1)]
        self.feats, self.dims = feats, dims
        # the first convolution: C == 1 with a 3x3x3 kernel has its own kernels (taps are the contraction); any other input
        # (several modalities, an anisotropic first kernel) becomes a channels-last fp16 image zero-padded to a multiple of 16
        # channels and runs as an ordinary block (the weight panel pads the same channels with zeros)
        self.c1_path = in_channels == 1 and kernels[0] == (3, 3, 3)
        self.cin_pad = in_channels if in_channels % 8 == 0 else -(-in_channels // 16) * 16

        # ---- concat buffers (one per decoder level u; level u sits at encoder depth d = num_pool-1-u).
        # torch.cat((up, skip), 1) (generic_ViT_UNet.py:263) is never executed: producers write straight into the two
        # channel ranges of one (N,D,H,W,2c) buffer -- except where a part is 32 channels = 64 bytes per voxel (the top
        # level): interleaved, every 16-channel chunk step of the consumer touches all 128-byte positions and an XCD's
        # L2 (32 CUs x 128 KB of lines) thrashes (5.9 GB fetched for a 1.26 GB input, profiles/r01_pmc_traffic.json);
        # there the two parts stay SEPARATE tensors and the decoder conv takes both (lnn_conv3d_*_cat).
        self.cat, self.gcat, self.split_cat = [], [], []
        for u in range(num_pool):
            d = num_pool - 1 - u
            # (the two-tensor form exists in the specialised 3x3x3 kernels only; the decoder stage u convolves with kernel -(u + 1))
            split = feats[d] % 32 == 0 and feats[d] * 2 <= 64 and os.environ.get("LNN_NO_SPLIT_CAT", "0") != "1" and \
                kernels[-(u + 1)] == (3, 3, 3)
            self.split_cat.append(split)
            if split:
                self.cat.append((_cl(N, dims[d], feats[d], dev), _cl(N, dims[d], feats[d], dev)))
                self.gcat.append((_cl(N, dims[d], feats[d], dev), _cl(N, dims[d], feats[d], dev)))
            else:
                self.cat.append(_cl(N, dims[d], 2 * feats[d], dev))
                self.gcat.append(_cl(N, dims[d], 2 * feats[d], dev))

        self.image = torch.zeros((N,) + dims[0] + (() if self.c1_path else (self.cin_pad,)), dtype=torch.float16, device=dev)
        self.blocks: List[ConvBlock] = []
        self.ups: List[UpBlock] = []
        self.segs: List[SegHead] = []
        order: List[object] = []     # forward execution order

        def new_block(prefix, cin, cout, strides, kernel, x, gx, gx_acc, z_target, gz_target, in_dims):
            strides, kernel = tuple(strides), tuple(kernel)
            od = tuple((s - 1) // st + 1 for s, st in zip(in_dims, strides))
            iso = kernel == (3, 3, 3) and strides in ((1, 1, 1), (2, 2, 2))
            blk = ConvBlock(prefix, cin, cout, strides[0] if iso else 0, kernel=kernel, strides=strides, cin_k=cin, x=x, gx=gx,
                            gx_accumulate=gx_acc, in_dims=in_dims)
            blk.y = _cl(N, od, cout, dev)
            if z_target is None:
                zb, gzb = _cl(N, od, cout, dev), _cl(N, od, cout, dev)
                blk.z, blk.gz = Act(zb, 0, cout), Act(gzb, 0, cout)
            else:
                blk.z, blk.gz = z_target, gz_target
            blk.mean = torch.zeros(N * cout, device=dev)
            blk.rstd = torch.zeros(N * cout, device=dev)
            blk.w = arena.by_name[prefix + ".conv.weight"]
            blk.b = arena.by_name[prefix + ".conv.bias"]
            blk.gamma = arena.by_name[prefix + ".instnorm.weight"]
            blk.beta = arena.by_name[prefix + ".instnorm.bias"]
            assert blk.w.shape == (cout, cin) + kernel, f"{prefix}: parameter arena built for another plan"
            self.blocks.append(blk)
            order.append(blk)
            return blk

        # ---- encoder
        one = (1, 1, 1)
        x, gx = (None, None) if self.c1_path else (Act(self.image, 0, self.cin_pad), None)
        cin = in_channels
        for d in range(num_pool):
            u = num_pool - 1 - d
            if self.split_cat[u]:
                skip, gskip = Act(self.cat[u][1], 0, feats[d]), Act(self.gcat[u][1], 0, feats[d])
            else:
                skip = Act(self.cat[u], feats[d], feats[d])
                gskip = Act(self.gcat[u], feats[d], feats[d])
            in_dims = dims[d - 1] if d > 0 else dims[0]
            b0 = new_block(f"conv_blocks_context.{d}.blocks.0", cin, feats[d], pools[d - 1] if d > 0 else one, kernels[d],
                           x, gx, d > 0, None, None, in_dims)
            if d == 0:
                b0.first = True
                b0.cin_k = 1 if self.c1_path else self.cin_pad
            b1 = new_block(f"conv_blocks_context.{d}.blocks.1", feats[d], feats[d], one, kernels[d], b0.z, b0.gz, False,
                           skip, gskip, dims[d])
            b1.x_block = b0
            x, gx, cin = b1.z, b1.gz, feats[d]
        # ---- bottleneck: Sequential(Stacked(1 strided conv), Stacked(1 conv))  (test_MultiHead_Module.py:394-415)
        nb = num_pool
        b0 = new_block(f"conv_blocks_context.{nb}.0.blocks.0", cin, feats[nb], pools[nb - 1], kernels[nb], x, gx, True, None, None,
                       dims[nb - 1])
        b1 = new_block(f"conv_blocks_context.{nb}.1.blocks.0", feats[nb], feats[nb], one, kernels[nb], b0.z, b0.gz, False, None, None,
                       dims[nb])
        b1.x_block = b0
        x, gx, cdown = b1.z, b1.gz, feats[nb]
        # ---- decoder
        for u in range(num_pool):
            d = num_pool - 1 - u
            cs = feats[d]
            if self.split_cat[u]:
                up_y, up_gy = Act(self.cat[u][0], 0, cs), Act(self.gcat[u][0], 0, cs)
                cat_act, gcat_act = up_y, up_gy
            else:
                up_y, up_gy = Act(self.cat[u], 0, cs), Act(self.gcat[u], 0, cs)
                cat_act, gcat_act = Act(self.cat[u], 0, 2 * cs), Act(self.gcat[u], 0, 2 * cs)
            up = UpBlock(f"tu.{u}", cdown, cs, strides=pools[-(u + 1)], x=x, gx=gx, y=up_y, gy=up_gy)
            up.w = arena.by_name[f"tu.{u}.weight"]
            assert up.w.shape == (cdown, cs) + pools[-(u + 1)], f"tu.{u}: parameter arena built for another plan"
            self.ups.append(up)
            order.append(up)
            b0 = new_block(f"conv_blocks_localization.{u}.0.blocks.0", 2 * cs, cs, one, kernels[-(u + 1)], cat_act, gcat_act, False,
                           None, None, dims[d])
            if self.split_cat[u]:
                b0.x2, b0.gx2 = Act(self.cat[u][1], 0, cs), Act(self.gcat[u][1], 0, cs)
            b1 = new_block(f"conv_blocks_localization.{u}.1.blocks.0", cs, cs, one, kernels[-(u + 1)], b0.z, b0.gz, False, None, None,
                           dims[d])
            b1.x_block = b0
            seg = SegHead(f"seg_outputs.{u}", cs, x=b1.z, gx=b1.gz, gx_has_prior=(u < num_pool - 1))
            seg.x_block = b1
            seg.w = arena.by_name[f"seg_outputs.{u}.weight"]
            self.segs.append(seg)
            order.append(seg)
            x, gx, cdown = b1.z, b1.gz, cs
        self.order = order
        # decoder blocks whose output feeds a seg head directly (the head follows its block in execution order)
        self._seg_after = {id(seg.x_block): seg for seg in self.segs}

        self.theta, self.grad = arena.theta, arena.grad

        # ---- fp16 weight panels + fp32 wgrad panels
        wp_off, pn_off = 0, 0
        for item in order:
            if isinstance(item, ConvBlock):
                nt = item.ntaps
                if item.cin_k == 1:
                    item.wp_fwd = wp_off; wp_off += nat.query("lnn_packed_weight_elems", 1, item.cout, 27)
                    item.panel = pn_off; pn_off += nat.query("lnn_wgrad_panel_elems", 1, item.cout, 27)
                else:
                    item.wp_fwd = wp_off; wp_off += nat.query("lnn_packed_weight_elems", nt, item.cout, item.cin)
                    item.wp_dgrad = wp_off; wp_off += nat.query("lnn_packed_weight_elems", nt, item.cin, item.cout)
                    # The weight gradient of a [1,3,3] stride-1 convolution IS the kz = 1 slice of the 3x3x3 weight gradient of the same
                    # tensors (dW[kz = 1, ky, kx] = sum_v dy[v] x[v + (0, ky - 1, kx - 1)]): the stride-1 tile kernel (~1000 TFLOP/s,
                    # two thirds of it on taps nobody reads) beats the flattened-voxel kernel (204-234 TFLOP/s) on the first stages of
                    # an anisotropic plan; fp32 panels are tap-major, so the 9 wanted taps are one contiguous run of a 27-tap panel.
                    # LNN_K133_WGRAD_333=0: the generic-geometry weight gradient (A/B switch)
                    item.wgrad_333 = (item.kernel == (1, 3, 3) and item.strides == (1, 1, 1) and item.cin_k % 8 == 0
                                      and os.environ.get("LNN_K133_WGRAD_333", "1") != "0")
                    item.panel = pn_off; pn_off += nat.query("lnn_wgrad_panel_elems", 27 if item.wgrad_333 else nt, item.cout, item.cin)
            elif isinstance(item, UpBlock):
                nt = item.ntaps
                item.wp_fwd = wp_off; wp_off += nat.query("lnn_packed_weight_elems", nt, item.cout, item.cin)
                item.wp_dgrad = wp_off; wp_off += nat.query("lnn_packed_weight_elems", nt, item.cin, item.cout)
                item.panel = pn_off; pn_off += nat.query("lnn_wgrad_panel_elems", nt, item.cin, item.cout)
            wp_off = (wp_off + 7) // 8 * 8
            pn_off = (pn_off + 3) // 4 * 4
        self.wpanels = torch.zeros(wp_off, dtype=torch.float16, device=dev)
        self.gpanels = torch.zeros(pn_off, device=dev)
        # descriptor tables of the batched pack / unpack launches (lnn_hip.h): one row per panel
        #   {src_off, dst_off, stride_m, stride_kc, stride_t, ntaps, M, KC, first}
        pk, up = [], []
        pk_total = up_total = 0

        def add_pack(w, dst, ntaps, M, KC, sm, skc, st):
            nonlocal pk_total
            pk.append([w.offset, dst, sm, skc, st, ntaps, M, KC, pk_total])
            pk_total += nat.query("lnn_packed_weight_elems", ntaps, M, KC)

        def add_unpack(w, panel, ntaps, M, KC, sm, skc, st):
            nonlocal up_total
            up.append([panel, w.offset, sm, skc, st, ntaps, M, KC, up_total])
            up_total += ntaps * M * KC

        for item in order:
            if isinstance(item, ConvBlock):
                K, C, nt = item.cout, item.cin, item.ntaps
                if item.cin_k == 1:
                    add_pack(item.w, item.wp_fwd, 1, K, 27, 27, 1, 0)
                    add_unpack(item.w, item.panel, 1, K, 27, 27, 1, 0)
                else:
                    add_pack(item.w, item.wp_fwd, nt, K, C, C * nt, nt, 1)
                    add_pack(item.w, item.wp_dgrad, nt, C, K, nt, C * nt, 1)
                    add_unpack(item.w, self._panel_live(item), nt, K, C, C * nt, nt, 1)
            elif isinstance(item, UpBlock):
                C, K, nt = item.cin, item.cout, item.ntaps
                add_pack(item.w, item.wp_fwd, nt, K, C, nt, K * nt, 1)
                add_pack(item.w, item.wp_dgrad, nt, C, K, K * nt, nt, 1)
                add_unpack(item.w, item.panel, nt, C, K, K * nt, nt, 1)
        self._pack_desc = torch.tensor(pk, dtype=torch.int64, device=dev)
        self._unpack_desc = torch.tensor(up, dtype=torch.int64, device=dev)
        self._pack_total, self._unpack_total = pk_total, up_total
        cmax = max(2 * f for f in feats)
        ws_doubles = max(nat.query("lnn_instnorm_ws_doubles", N, cmax), (nat.query("lnn_seg1x1_bwd_ws_floats", N, cmax) + 1) // 2,
                         nat.query("lnn_instnorm_lrelu_seg_bwd_ws_doubles", N, max(seg.cin for seg in self.segs)), 64)
        self.ws = torch.zeros(ws_doubles, dtype=torch.float64, device=dev)
        # fp32 split-K scratch of the small deep layers (see lnn_conv3d_dgrad_ws): 8 slices of the largest
        # [N][voxels][roundup32(channels)] among the stride-1 layers with fewer than 512 (8x8x8 x 32-channel) units; the
        # flattened-voxel kernels that take the volumes of <= 4096 output voxels (csrc/igemm_gen.hip) split the contraction up to
        # 64 ways: room for min(64, 2048 waves / their unsplit wave count) slices of every such output
        need = 1
        for blk in self.blocks:
            if blk.stride != 1 or blk.cin_k == 1:
                continue
            od = blk.in_dims
            vox = N * od[0] * od[1] * od[2]
            tiles = N * -(-od[0] // 8) * -(-od[1] // 8) * -(-od[2] // 8)
            for ch in (blk.cout, blk.cin):
                if tiles * -(-ch // 32) < 512:
                    need = max(need, 8 * vox * (-(-ch // 32) * 32))

        def gen_need(vox, ch, classes=1, always=False, limit=4096):
            # ``limit``: the volume up to which the isotropic entry points hand this op to the generic kernels (lnn_gen_prefers,
            # csrc/igemm_conv.hip: stride-1 / stride-2 convolutions <= 4096 voxels, transposed-conv data gradients <= 20000)
            if vox > limit and not always:                # (isotropic layers above the threshold stay on the specialised kernels)
                return 1
            mp = -(-ch // 32) * 32
            waves = classes * -(-(vox // classes) // 64) * -(-mp // 64)
            ks = 1
            while waves * ks * 2 <= 2048 and ks < 64:
                ks *= 2
            return ks * vox * mp if ks > 1 else 1
        for blk in self.blocks:
            if blk.cin_k == 1:
                continue
            ncls = blk.strides[0] * blk.strides[1] * blk.strides[2]
            need = max(need, gen_need(N * blk.z.V, blk.cout, 1, not blk.iso))                                     # forward
            need = max(need, gen_need(N * blk.in_dims[0] * blk.in_dims[1] * blk.in_dims[2], blk.cin_k, ncls, not blk.iso))   # data gradient
        for up in self.ups:
            need = max(need, gen_need(N * up.y.V, up.cout, up.ntaps, not up.iso), gen_need(N * up.x.V, up.cin, 1, not up.iso, 20000))
        self.splitk_ws = torch.zeros(need, dtype=torch.float32, device=dev)
        self.packed_version = -1
        self.unused_heads: List[str] = []
        self._sides = {}
        self.overlap_wgrad = os.environ.get("LNN_NO_WGRAD_OVERLAP", "0") != "1"
        # the conv-bias gradient in front of an InstanceNorm is sum_v dy = 0 analytically; True sums the fp16 rounding noise of dy
        # the way autograd does (one more block reduction + launch per layer).  Either way the optimiser steps the bias
        # (weight decay, momentum) exactly as torch does with a ~0 gradient.
        self.deterministic_wgrad = os.environ.get("LNN_DETERMINISTIC_WGRAD", "0") == "1"
        self._det_buf = None
        self.fuse_seg_fwd = os.environ.get("LNN_NO_FUSED_SEG", "0") != "1"           # A/B switch (measurements only)
        self.numeric_conv_bias_grad = os.environ.get("LNN_NUMERIC_CONV_BIAS_GRAD", "0") == "1"
        self.fuse_in_stats = os.environ.get("LNN_NO_FUSED_IN_STATS", "0") != "1"     # A/B switch (measurements only)
        # seg head backward folded into the InstanceNorm backward of the block that feeds it (dL/dz never written): A/B switch
        self.fuse_seg_bwd = os.environ.get("LNN_NO_FUSED_SEG_BWD", "0") != "1"
        # first block: pass 2 of its InstanceNorm backward rebuilt inside its weight gradient (dy never written): A/B switch
        self.fuse_first_bwd = os.environ.get("LNN_NO_FUSED_FIRST_BWD", "0") != "1"
        # second block of a stage: pass 1 of the FIRST block's InstanceNorm backward rides the data gradient that produces its dL/dz
        # (lnn_conv3d_dgrad_in_bwd_sums; fused inside the z-streaming kernel where an instance exists): A/B switch
        self.fuse_in_bwd_reduce = os.environ.get("LNN_NO_FUSED_IN_BWD_REDUCE", "0") != "1"
        # volumes up to this many voxels per sample run their normalisation as ONE launch per direction (csrc/norm_act.hip
        # in_small_*; LNN_IN_SMALL=0: the multi-launch passes everywhere, A/B switch)
        self.small_v = nat.query("lnn_instnorm_small_volume") if os.environ.get("LNN_IN_SMALL", "1") != "0" else 0
        # measurement hook (bench.py): {"layer": <block prefix>} -> the forward conv / data-gradient / weight-gradient calls of
        # that block are bracketed with timing events ON THE STREAM THEY LAUNCH ON, appended to probe["fwd" | "dgrad" | "wgrad"]
        self.probe = None
        # arena offset right after each item's parameter slots (= start of the next item's slots)
        self._watermark = {}
        for item in order:
            slots = [s for s in (getattr(item, a, None) for a in ("w", "b", "gamma", "beta")) if s is not None]
            self._watermark[id(item)] = max(s.offset + (s.numel + 3) // 4 * 4 for s in slots)

    def _probed(self, kind, item, fn):
        pr = self.probe
        if pr is None or (pr.get("layer") != item.prefix and pr.get("layer") != "*"):
            fn()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        if pr.get("layer") == "*":          # tools/layer_table.py: every call of every layer
            pr.setdefault("all", []).append((item.prefix, kind, e0, e1))
        else:
            pr.setdefault(kind, []).append((e0, e1))

    # ------------------------------------------------------------------------------------------ views
    def pview(self, slot: ParamSlot, arena=None):
        """View of a parameter slot in theta (default) or another flat arena; memoised: a step asks ~600 times and a fresh
        slice + view costs 4 us of host time each (the arenas never move)."""
        a = self.theta if arena is None else arena
        key = (slot.name, a.data_ptr())
        v = self._pviews.get(key)
        if v is None:
            v = self._pviews[key] = a[slot.offset:slot.offset + slot.numel].view(slot.shape)
        return v

    @staticmethod
    def _panel_live(item):
        """Offset (in floats) of the taps of ``item``'s weight-gradient panel that are folded into the gradient: the whole panel, or
        the kz = 1 run of the 27-tap panel a [1,3,3] block computes (see ``wgrad_333``)."""
        if getattr(item, "wgrad_333", False):
            return item.panel + 9 * (-(-item.cout // 32) * 32) * (-(-item.cin // 32) * 32)
        return item.panel

    def _wp(self, off):
        return _Ptr(self.wpanels, off)

    def _pn(self, off):
        return _Ptr(self.gpanels, off)

    # ------------------------------------------------------------------------------------------ pack
    def pack_weights(self):
        """fp32 parameter arena -> fp16 MFMA panels of every layer, one launch.  (Round 5 measured the re-pack of all panels but the
        first block's on a side stream next to the first block's kernels: the 98-us pack and the C = 1 convolution are both
        HBM-bound and slowed each other down -- 119 + 357 us side by side against 98 + 190 in sequence; not kept.)"""
        nat.call("lnn_pack_weights_batched", self.theta, self.wpanels, self._pack_desc, self._pack_desc.shape[0],
                 self._pack_total)
        self.packed_version = self.arena.version

    def unpack_wgrads(self):
        """fp32 wgrad panels of every layer += into the gradient arena (PyTorch layouts), one launch."""
        nat.call("lnn_unpack_wgrad_batched", self.gpanels, self.grad, self._unpack_desc, self._unpack_desc.shape[0],
                 self._unpack_total, 1.0, 1)

    @staticmethod
    def _at(obj, n0):
        """Pointer to sample n0 of an activation (Act) or batch-major tensor."""
        if isinstance(obj, Act):
            return _Ptr(obj.buf, n0 * obj.V * obj.ld + obj.off)
        return _Ptr(obj, n0 * obj.stride(0)) if n0 else obj

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor, seg_weights: Optional[List[torch.Tensor]] = None, body: bool = True):
        """x: (N,1,D,H,W) float32 on the device.  Returns logits per decoder level u (low-res first, as
        ``seg_outputs`` is indexed upstream), fp32 (N,K,d,h,w).  ``seg_weights`` overrides the seg-head
        parameters (used to evaluate several heads on one body pass); ``body=False`` reuses the stored
        body activations."""
        N = self.N
        assert tuple(x.shape) == (N, self.in_channels) + self.patch, \
            f"engine built for {(N, self.in_channels) + self.patch}, got {tuple(x.shape)}"
        if self.packed_version != self.arena.version:
            self.pack_weights()
        if body:
            if self.c1_path:
                nat.call("lnn_cast_f32_to_h", x.contiguous(), self.image, x.numel())
            else:       # (N, C, D, H, W) fp32 -> channels-last fp16, channels [C, cin_pad) stay zero
                nat.call("lnn_image_to_cl_h", x.contiguous(), self.image, N, self.in_channels, x[0, 0].numel(), self.cin_pad)
        logits = [torch.empty((N, self.K) + seg.x.dims, device=self.device) for seg in self.segs]
        sw = None if seg_weights is None else [w.contiguous() for w in seg_weights]
        ws, splitk_ws = self.ws, self.splitk_ws
        u = 0
        fused_segs = set()
        for item in self.order:
            if isinstance(item, ConvBlock):
                if not body:
                    continue
                D, H, W = item.in_dims
                xin = self.image if item.x is None else item.x
                ldx = 1 if item.x is None else item.x.ld
                C = item.cout
                V = item.z.V
                mean, rstd = item.mean, item.rstd
                seg = self._seg_after.get(id(item))
                seg_fused = seg is not None and sw is None and self.fuse_seg_fwd and (C // 8) & (C // 8 - 1) == 0 and C <= 512
                if item.iso and self.fuse_in_stats and V <= self.small_v and not seg_fused:
                    # the lowest levels (<= 2048 voxels per sample): statistics, their finalize and the normalisation are one launch
                    # behind the convolution instead of three (lnn_conv3d_fwd_in_lrelu)
                    self._probed("fwd", item, lambda: nat.call(
                        "lnn_conv3d_fwd_in_lrelu", xin, item.x2, ldx, item.x.C if item.x2 is not None else 0,
                        self._wp(item.wp_fwd), self.pview(item.b), item.y, N, D, H, W, item.cin_k, C, item.stride, IN_EPS, mean, rstd,
                        self.pview(item.gamma), self.pview(item.beta), LRELU_SLOPE, item.z, item.z.ld, ws, splitk_ws,
                        splitk_ws.numel()))
                    continue
                if not item.iso:
                    # per-axis kernel / stride from the plans: generic-geometry kernel, statistics as a separate pass
                    self._probed("fwd", item, lambda: nat.call(
                        "lnn_conv3d_fwd_g", xin, ldx, self._wp(item.wp_fwd), self.pview(item.b), item.y, C, N, D, H, W,
                        item.cin_k, C, *item.kernel, *item.strides, splitk_ws, splitk_ws.numel()))
                    nat.call("lnn_instnorm_stats", item.y, N, V, C, IN_EPS, mean, rstd, ws)
                elif self.fuse_in_stats:
                    # conv + InstanceNorm statistics in one call (the z-streaming kernel sums in its epilogue)
                    self._probed("fwd", item, lambda: nat.call(
                        "lnn_conv3d_fwd_in_stats", xin, item.x2, ldx,
                        item.x.C if item.x2 is not None else 0, self._wp(item.wp_fwd), self.pview(item.b),
                        item.y, N, D, H, W, item.cin_k, C, item.stride, IN_EPS, mean, rstd, ws,
                        splitk_ws, splitk_ws.numel()))
                else:
                    if item.x2 is not None:
                        nat.call("lnn_conv3d_fwd_cat", xin, item.x2, ldx, item.x.C, self._wp(item.wp_fwd),
                                 self.pview(item.b), item.y, C, N, D, H, W, item.cin_k, C)
                    else:
                        nat.call("lnn_conv3d_fwd", xin, ldx, self._wp(item.wp_fwd), self.pview(item.b), item.y, C,
                                 N, D, H, W, item.cin_k, C, item.stride)
                    nat.call("lnn_instnorm_stats", item.y, N, V, C, IN_EPS, mean, rstd, ws)
                if seg_fused:
                    # decoder block that feeds a seg head: InstanceNorm + LeakyReLU + the 1x1x1 head in one pass over y
                    # (measured in round 5: NOT writing the last block's normalised tensor -- nobody reads it in a plain training
                    # step -- changes nothing, 20.01 vs 20.02 / 19.94 ms: the pass is bound by its instruction stream, not by its
                    # stores, and LwF's old heads do read it)
                    self._probed("in_fwd", item, lambda: nat.call(
                        "lnn_instnorm_lrelu_seg_fwd", item.y, item.z, item.z.ld, N, V, C, mean, rstd,
                        self.pview(item.gamma), self.pview(item.beta), LRELU_SLOPE, self.pview(seg.w),
                        logits[self.segs.index(seg)], self.K))
                    fused_segs.add(id(seg))
                else:
                    self._probed("in_fwd", item, lambda: nat.call(
                        "lnn_instnorm_lrelu_fwd", item.y, item.z, item.z.ld, N, V, C, mean, rstd,
                        self.pview(item.gamma), self.pview(item.beta), LRELU_SLOPE))
            elif isinstance(item, UpBlock):
                if not body:
                    continue
                D, H, W = item.x.dims
                if item.iso:
                    self._probed("fwd", item, lambda: nat.call(
                        "lnn_convT3d_k2s2_fwd_ws", item.x, item.x.ld, self._wp(item.wp_fwd), item.y,
                        item.y.ld, N, D, H, W, item.cin, item.cout, splitk_ws, splitk_ws.numel()))
                else:
                    self._probed("fwd", item, lambda: nat.call(
                        "lnn_convT3d_fwd_g", item.x, item.x.ld, self._wp(item.wp_fwd), item.y,
                        item.y.ld, N, D, H, W, item.cin, item.cout, *item.strides, splitk_ws, splitk_ws.numel()))
            else:
                if id(item) not in fused_segs:
                    w = self.pview(item.w) if sw is None else sw[u]
                    nat.call("lnn_seg1x1_fwd", item.x, item.x.ld, w, logits[u], N, item.x.V, item.cin, self.K)
                u += 1
        return logits

    def conv_outputs(self, logits):
        """name -> (N,C,D,H,W) strided VIEW of the output of every conv / transposed conv / seg head of the LAST forward,
        in execution order -- what the reference's forward hooks on every ``conv.Conv*`` module record
        (plop/nnUNetTrainerPLOP.py:335-358).  ``logits``: the list that forward returned (low resolution first).  The views
        alias the engine's buffers: read them before the next forward / backward."""
        from collections import OrderedDict
        out, u = OrderedDict(), 0
        for item in self.order:
            if isinstance(item, ConvBlock):
                out[item.prefix + ".conv"] = item.y.permute(0, 4, 1, 2, 3)
            elif isinstance(item, UpBlock):
                out[item.prefix] = item.y.tensor().permute(0, 4, 1, 2, 3)
            else:
                out[item.prefix] = logits[u]
                u += 1
        return out

    def _det_scratch(self):
        """fp32 scratch of the deterministic weight gradients (``deterministic_wgrad`` / LNN_DETERMINISTIC_WGRAD=1): every writer
        of a weight-gradient block stores into its own panel copy, an ordered reduction replaces the fp32 atomics -> two
        identical steps give bit-identical parameters.  64 M floats cover every layer of the BASELINE configs (the library
        reports an error, not a fallback, if a layer needs more).  All weight gradients run on ONE stream, so one scratch."""
        if not self.deterministic_wgrad:
            return None
        if self._det_buf is None:
            self._det_buf = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=self.device)
        return self._det_buf

    # ------------------------------------------------------------------------------------------ backward
    def backward(self, dlogits: List[Optional[torch.Tensor]], skip_body: bool = False, progress=None):
        """dlogits[u]: gradient wrt ``logits[u]`` (fp32, already carrying the loss scale) or None.
        Accumulates parameter gradients into the flat arena ``self.grad`` (scaled like dlogits)."""
        N = self.N
        self.gpanels.zero_()
        self.unused_heads = [seg.w.name for seg, dl in zip(self.segs, dlogits) if dl is None]
        dls = [None if dl is None else dl.contiguous() for dl in dlogits]
        ws, splitk_ws = self.ws, self.splitk_ws
        # Weight gradients are off the critical path of backward (they only have to be final before the optimiser /
        # the all-reduce): they go to a SIDE HIP stream right after the layer's dL/dy exists.
        main = torch.cuda.current_stream()
        # data parallel (progress given): the weight gradients stay on the side stream; a layer's panel is folded into the
        # gradient arena there too, and the all-reduce of every bucket that became final is launched FROM the side stream
        # (parallel.GradAllReducer.progress): two streams share the chip, as in the single-GPU plan
        side = self._side_stream() if self.overlap_wgrad else None
        def on_side(fn):
            # (round 5 also measured the first layer's HBM-bound weight gradient on a stream of its own, next to the second block's
            # MFMA-bound one instead of behind it: the trace's tail shrank by 0.17 ms, the step by 0.00-0.04 ms -- the two kernels
            # slow each other down; not kept, profiles/r05_switches_ab.txt)
            if side is None:
                fn()
                return
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                fn()

        # the DP all-reduce overlaps with backward and needs every layer's gradient final as soon as its wgrad is:
        # per-layer unpack there; otherwise one batched unpack after the last wgrad
        per_layer_unpack = progress is not None

        def unpack(item):
            nt = item.ntaps
            if isinstance(item, ConvBlock):
                K, C = item.cout, item.cin
                gw = self.pview(item.w, self.grad)
                if item.cin_k == 1:
                    nat.call("lnn_unpack_wgrad", self._pn(item.panel), gw, 1, K, 27, 27, 1, 0, 1.0, 1)
                else:
                    nat.call("lnn_unpack_wgrad", self._pn(self._panel_live(item)), gw, nt, K, C, C * nt, nt, 1, 1.0, 1)
            else:
                C, K = item.cin, item.cout
                nat.call("lnn_unpack_wgrad", self._pn(item.panel), self.pview(item.w, self.grad), nt, C, K, K * nt, nt, 1, 1.0, 1)

        fuse_seg = self.fuse_seg_bwd and not skip_body and not self.numeric_conv_bias_grad and self.K <= 4
        seg_u = len(self.segs)
        pending = {}          # id(block) -> (seg head, dlogits): heads whose backward runs inside their block's norm backward
        presummed = set()     # id(block): pass 1 of its normalisation backward was taken by the data gradient that produced its dL/dz
        normed = set()        # id(block): its WHOLE normalisation backward was (small volumes, lnn_conv3d_dgrad_in_bwd)
        for item in reversed(self.order):
            if progress is not None and item is not self.order[-1]:
                # everything after this item in the arena is final once main (norm / bias / seg gradients) and side
                # (weight gradients + their unpack) have run what is enqueued so far; a head that was deferred into
                # this block's call is not final yet: only what lies behind the head is
                wm_item = pending[id(item)][0] if id(item) in pending else item
                progress(self._watermark[id(wm_item)], side)
            if isinstance(item, SegHead):
                seg_u -= 1
                dl = dls[seg_u]
                if dl is None:
                    if not item.gx_has_prior:
                        item.gx.buf.zero_()
                    continue
                if fuse_seg:
                    pending[id(item.x_block)] = (item, dl)
                    continue
                gw = self.pview(item.w, self.grad).view(self.K, item.cin)
                nat.call("lnn_seg1x1_bwd", item.x, item.x.ld, self.pview(item.w), dl, item.gx,
                         item.gx.ld, gw, N, item.x.V, item.cin, self.K, 1 if item.gx_has_prior else 0, 1.0, ws)
            elif skip_body:
                continue
            elif isinstance(item, ConvBlock):
                V, K, C = item.z.V, item.cout, item.cin_k
                if id(item) in pending:
                    seg, dl = pending.pop(id(item))
                    self._probed("in_bwd", item, lambda: nat.call(
                        "lnn_instnorm_lrelu_seg_bwd", item.y, item.gz if seg.gx_has_prior else None,
                        item.gz.ld, self.pview(seg.w), dl, self.pview(seg.w, self.grad).view(self.K, seg.cin), self.K,
                        N, V, K, item.mean, item.rstd, self.pview(item.gamma), self.pview(item.beta),
                        LRELU_SLOPE, self.pview(item.gamma, self.grad), self.pview(item.beta, self.grad), 1.0, ws))
                elif item.cin_k == 1 and self.fuse_first_bwd and not self.numeric_conv_bias_grad:
                    # the first block has no data gradient: only the sums of its normalisation backward are taken here (or were,
                    # by the data gradient of the block behind it), dy is rebuilt tile by tile inside the weight gradient below
                    # (lnn_conv3d_wgrad_c1_in_bwd)
                    if id(item) not in presummed:
                        self._probed("in_bwd", item, lambda: nat.call(
                            "lnn_instnorm_lrelu_bwd_sums", item.y, item.gz, item.gz.ld, N, V, K,
                            item.mean, item.rstd, self.pview(item.gamma), self.pview(item.beta),
                            LRELU_SLOPE, self.pview(item.gamma, self.grad), self.pview(item.beta, self.grad), 1.0, ws))
                    D, H, W = item.in_dims

                    def first_wgrad(item=item, K=K, D=D, H=H, W=W):
                        det = self._det_scratch()
                        nat.call("lnn_conv3d_wgrad_c1_in_bwd", self.image, item.y, item.gz, item.gz.ld,
                                 self._pn(item.panel), N, D, H, W, K, item.mean, item.rstd,
                                 self.pview(item.gamma), self.pview(item.beta), LRELU_SLOPE, ws, det,
                                 0 if det is None else det.numel())
                        if per_layer_unpack:
                            unpack(item)
                    on_side(lambda: self._probed("wgrad", item, first_wgrad))
                    continue
                elif id(item) in normed:
                    pass
                elif id(item) in presummed:
                    self._probed("in_bwd", item, lambda: nat.call(
                        "lnn_instnorm_lrelu_bwd_apply", item.y, item.gz, item.gz.ld, N, V, K,
                        item.mean, item.rstd, self.pview(item.gamma), self.pview(item.beta), LRELU_SLOPE, ws))
                else:
                    self._probed("in_bwd", item, lambda: nat.call(
                        "lnn_instnorm_lrelu_bwd", item.y, item.gz, item.gz.ld, N, V, K,
                        item.mean, item.rstd, self.pview(item.gamma), self.pview(item.beta), LRELU_SLOPE,
                        self.pview(item.gamma, self.grad), self.pview(item.beta, self.grad),
                        self.pview(item.b, self.grad) if self.numeric_conv_bias_grad else None, 1.0, ws))
                D, H, W = item.in_dims
                xin = self.image if item.x is None else item.x
                ldx = 1 if item.x is None else item.x.ld

                def conv_wgrad(item=item, xin=xin, ldx=ldx, K=K, C=C, D=D, H=H, W=W):
                    self._probed("wgrad", item, lambda: conv_wgrad_call(item, xin, ldx, K, C, D, H, W))
                    if per_layer_unpack:
                        unpack(item)

                def conv_wgrad_call(item, xin, ldx, K, C, D, H, W):
                    det = self._det_scratch()
                    if item.wgrad_333:
                        # the 27-tap panel of this block; only its taps 9..17 (kz = 1) are folded into the gradient
                        if det is not None:
                            nat.call("lnn_conv3d_wgrad_det", xin, ldx, item.y, K, self._pn(item.panel), N, D, H, W, C, K, 1, det, det.numel())
                        else:
                            nat.call("lnn_conv3d_wgrad", xin, ldx, item.y, K, self._pn(item.panel), N, D, H, W, C, K, 1)
                    elif not item.iso:
                        nat.call("lnn_conv3d_wgrad_g", xin, ldx, item.y, K, self._pn(item.panel), N, D, H, W, C, K,
                                 *item.kernel, *item.strides, det, 0 if det is None else det.numel())
                    elif item.x2 is not None and det is not None:
                        nat.call("lnn_conv3d_wgrad_cat_det", xin, item.x2, ldx, item.x.C, item.y, K,
                                 self._pn(item.panel), N, D, H, W, C, K, det, det.numel())
                    elif item.x2 is not None:
                        nat.call("lnn_conv3d_wgrad_cat", xin, item.x2, ldx, item.x.C, item.y, K,
                                 self._pn(item.panel), N, D, H, W, C, K)
                    elif det is not None:
                        nat.call("lnn_conv3d_wgrad_det", xin, ldx, item.y, K, self._pn(item.panel), N, D, H, W, C, K,
                                 item.stride, det, det.numel())
                    else:
                        nat.call("lnn_conv3d_wgrad", xin, ldx, item.y, K, self._pn(item.panel), N, D, H, W, C, K,
                                 item.stride)
                on_side(conv_wgrad)
                if item.first or item.gx is None:
                    pass                                   # the first convolution has no data gradient
                elif not item.iso:
                    self._probed("dgrad", item, lambda: nat.call(
                        "lnn_conv3d_dgrad_g", item.y, K, self._wp(item.wp_dgrad), item.gx, item.gx.ld,
                        N, D, H, W, C, K, *item.kernel, *item.strides, 1 if item.gx_accumulate else 0, splitk_ws,
                        splitk_ws.numel()))
                elif (self.fuse_in_bwd_reduce and item.x_block is not None and item.stride == 1 and item.gx2 is None and item.iso
                      and not item.gx_accumulate and not self.numeric_conv_bias_grad and item.x_block.z.V <= self.small_v
                      and not item.x_block.first):
                    # the lowest levels: the WHOLE normalisation backward of the stage's first block (reduce, sums, apply) is one launch
                    # behind the data gradient that produces its dL/dz (lnn_conv3d_dgrad_in_bwd); its turn in the loop skips the pass
                    xb = item.x_block
                    self._probed("dgrad", item, lambda: nat.call(
                        "lnn_conv3d_dgrad_in_bwd", item.y, K, self._wp(item.wp_dgrad), item.gx, item.gx.ld,
                        N, D, H, W, C, K, xb.y, xb.mean, xb.rstd, self.pview(xb.gamma),
                        self.pview(xb.beta), LRELU_SLOPE, self.pview(xb.gamma, self.grad), self.pview(xb.beta, self.grad), 1.0, ws,
                        splitk_ws, splitk_ws.numel()))
                    normed.add(id(xb))
                elif (self.fuse_in_bwd_reduce and item.x_block is not None and item.stride == 1 and item.gx2 is None
                      and not item.gx_accumulate and not self.numeric_conv_bias_grad):
                    # dL/dz of the stage's first block AND pass 1 of its normalisation backward (nothing between here and that
                    # block's turn in the loop uses ws)
                    xb = item.x_block
                    self._probed("dgrad", item, lambda: nat.call(
                        "lnn_conv3d_dgrad_in_bwd_sums", item.y, K, self._wp(item.wp_dgrad), item.gx, item.gx.ld,
                        N, D, H, W, C, K, xb.y, xb.mean, xb.rstd, self.pview(xb.gamma),
                        self.pview(xb.beta), LRELU_SLOPE, self.pview(xb.gamma, self.grad), self.pview(xb.beta, self.grad), 1.0, ws,
                        splitk_ws, splitk_ws.numel()))
                    presummed.add(id(xb))
                elif item.gx2 is not None:
                    self._probed("dgrad", item, lambda: nat.call(
                        "lnn_conv3d_dgrad_cat_ws", item.y, K, self._wp(item.wp_dgrad), item.gx,
                        item.gx2, item.gx.ld, item.gx.C, N, D, H, W, C, K, 1 if item.gx_accumulate else 0,
                        splitk_ws, splitk_ws.numel()))
                else:
                    self._probed("dgrad", item, lambda: nat.call(
                        "lnn_conv3d_dgrad_ws", item.y, K, self._wp(item.wp_dgrad), item.gx, item.gx.ld,
                        N, D, H, W, C, K, item.stride, 1 if item.gx_accumulate else 0, splitk_ws,
                        splitk_ws.numel()))
            else:  # UpBlock
                D, H, W = item.x.dims
                C, K = item.cin, item.cout

                def up_wgrad(item=item, C=C, K=K, D=D, H=H, W=W):
                    self._probed("wgrad", item, lambda: up_wgrad_call(item, C, K, D, H, W))
                    if per_layer_unpack:
                        unpack(item)

                def up_wgrad_call(item, C, K, D, H, W):
                    det = self._det_scratch()
                    if not item.iso:
                        nat.call("lnn_convT3d_wgrad_g", item.x, item.x.ld, item.gy, item.gy.ld,
                                 self._pn(item.panel), N, D, H, W, C, K, *item.strides, det, 0 if det is None else det.numel())
                    elif det is not None:
                        nat.call("lnn_convT3d_k2s2_wgrad_det", item.x, item.x.ld, item.gy, item.gy.ld,
                                 self._pn(item.panel), N, D, H, W, C, K, det, det.numel())
                    else:
                        nat.call("lnn_convT3d_k2s2_wgrad", item.x, item.x.ld, item.gy, item.gy.ld,
                                 self._pn(item.panel), N, D, H, W, C, K)
                on_side(up_wgrad)
                if item.iso:
                    self._probed("dgrad", item, lambda: nat.call(
                        "lnn_convT3d_k2s2_dgrad_ws", item.gy, item.gy.ld, self._wp(item.wp_dgrad), item.gx,
                        item.gx.ld, N, D, H, W, C, K, 0, splitk_ws, splitk_ws.numel()))
                else:
                    self._probed("dgrad", item, lambda: nat.call(
                        "lnn_convT3d_dgrad_g", item.gy, item.gy.ld, self._wp(item.wp_dgrad), item.gx,
                        item.gx.ld, N, D, H, W, C, K, *item.strides, 0, splitk_ws, splitk_ws.numel()))
        if progress is not None:
            progress(0, side)           # the first layer's gradients are enqueued: the last bucket leaves from here too
        if side is not None:
            main.wait_stream(side)      # every weight gradient is final before anything downstream (norm, step)
        if not per_layer_unpack and not skip_body:
            self.unpack_wgrads()

    def _side_stream(self, i=0):
        """Side HIP stream ``i`` of this engine's device (0 = weight gradients).  PROCESS-WIDE, shared by every engine: the runtime multiplexes
        streams onto a handful of hardware queues, and an engine whose side stream lands on the queue of its main stream loses
        the overlap it exists for (measured in round 5: the trainers built later in one process -- each engine with three streams
        of its own -- ran 1 ms per step slower)."""
        idx = self.device.index
        if idx is None:
            idx = torch.cuda.current_device() if self.device.type == "cuda" else -1
        key = (idx, i)
        st = _SIDE_STREAMS.get(key)
        if st is None:
            st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=self.device)
        self._sides[i] = st
        return st

    # ------------------------------------------------------------------------------------------ stats
    def flops_per_patch(self):
        """Algorithmic conv-stack FLOPs of one training patch: 6*MAC_fwd - 2*MAC(first layer)."""
        mac = 0
        first = 0
        for item in self.order:
            if isinstance(item, ConvBlock):
                m = item.z.V * item.cin * item.cout * item.ntaps
                mac += m
                if item.first:
                    first = m
            elif isinstance(item, UpBlock):
                mac += item.x.V * item.cin * item.cout * item.ntaps
            else:
                mac += item.x.V * item.cin * self.K
        return 6 * mac - 2 * first, mac


_SIDE_STREAMS = {}      # (device index, i) -> torch.cuda.Stream, see UNetEngine._side_stream


class _Ptr:
    """Element-offset pointer into a flat tensor (for the ctypes layer)."""

    def __init__(self, t, off):
        self.t, self.off = t, off

    def data_ptr(self):
        return self.t.data_ptr() + self.off * self.t.element_size()
