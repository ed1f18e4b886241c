"""fp32-STORAGE launch plan of the 3-D Generic_UNet: the reference's ``fp16=False`` branch
(multihead/nnUNetTrainerMultiHead.py:632-641; CLI ``--fp32`` run/run_training.py:71).

Same network, same parameter arena, same interface as ``engine.UNetEngine`` (``forward`` / ``backward`` /
``flops_per_patch``), but every activation and gradient buffer is fp32 and the kernels are the direct fp32 ones of
``csrc/ref_f32.hip`` (fp64 accumulation in a fixed order: bit-reproducible).  This is the PARITY mode: it follows the
reference's CPU/fp32 arithmetic to round-off, so Fisher values, Riemannian-Walk scores, update vectors and
multi-iteration loss curves can be asserted at 1e-4 and tighter (tests/test_fp32_parity_gpu.py); it is ~30x slower than the
fp16-storage MFMA plan and is not the path the benchmark measures.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import native as nat
from .engine import IN_EPS, LRELU_SLOPE, Act, ParamArena, unet_geometry


def _cl32(N, dims, C, device):
    return torch.zeros((N,) + tuple(dims) + (C,), dtype=torch.float32, device=device)


class _Blk:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class UNetEngineF32:
    storage = "fp32"

    def __init__(self, arena: ParamArena, in_channels, base_features, num_classes, num_pool, patch_size, batch_size,
                 device="cuda", max_features=320, pool_op_kernel_sizes=None, conv_kernel_sizes=None):
        self.arena = arena
        pools, kernels, dims = unet_geometry(num_pool, patch_size, pool_op_kernel_sizes, conv_kernel_sizes)
        self.pools, self.kernels = pools, kernels
        self.in_channels, self.K, self.num_pool = in_channels, num_classes, num_pool
        self.patch, self.N, self.device = tuple(patch_size), batch_size, torch.device(device)
        dev, N = self.device, batch_size
        feats = [min(base_features * 2 ** d, max_features) for d in range(num_pool + 1)]
        self.feats, self.dims = feats, dims
        self.cat = [_cl32(N, dims[num_pool - 1 - u], 2 * feats[num_pool - 1 - u], dev) for u in range(num_pool)]
        self.gcat = [torch.zeros_like(c) for c in self.cat]
        self.image = None
        self.order: List[_Blk] = []
        by = arena.by_name

        def block(prefix, cin, cout, strides, kernel, x, gx, gx_acc, z_target, gz_target, in_dims):
            od = tuple((s - 1) // st + 1 for s, st in zip(in_dims, strides))
            b = _Blk(kind="conv", prefix=prefix, cin=cin, cout=cout, strides=tuple(strides), kernel=tuple(kernel), x=x, gx=gx,
                     gx_accumulate=gx_acc, in_dims=in_dims,
                     y=_cl32(N, od, cout, dev), w=by[prefix + ".conv.weight"], b=by[prefix + ".conv.bias"],
                     gamma=by[prefix + ".instnorm.weight"], beta=by[prefix + ".instnorm.bias"],
                     mean=torch.zeros(N * cout, device=dev), rstd=torch.zeros(N * cout, device=dev))
            assert b.w.shape == (cout, cin) + tuple(kernel), f"{prefix}: parameter arena built for another plan"
            if z_target is None:
                b.z, b.gz = Act(_cl32(N, od, cout, dev), 0, cout), Act(_cl32(N, od, cout, dev), 0, cout)
            else:
                b.z, b.gz = z_target, gz_target
            self.order.append(b)
            return b

        one = (1, 1, 1)
        x, gx, cin = None, None, in_channels
        for d in range(num_pool):
            u = num_pool - 1 - d
            skip, gskip = Act(self.cat[u], feats[d], feats[d]), Act(self.gcat[u], feats[d], feats[d])
            b0 = block(f"conv_blocks_context.{d}.blocks.0", cin, feats[d], pools[d - 1] if d > 0 else one, kernels[d], x, gx, d > 0,
                       None, None, dims[d - 1] if d > 0 else dims[0])
            b1 = block(f"conv_blocks_context.{d}.blocks.1", feats[d], feats[d], one, kernels[d], b0.z, b0.gz, False, skip, gskip, dims[d])
            x, gx, cin = b1.z, b1.gz, feats[d]
        nb = num_pool
        b0 = block(f"conv_blocks_context.{nb}.0.blocks.0", cin, feats[nb], pools[nb - 1], kernels[nb], x, gx, True, None, None, dims[nb - 1])
        b1 = block(f"conv_blocks_context.{nb}.1.blocks.0", feats[nb], feats[nb], one, kernels[nb], b0.z, b0.gz, False, None, None, dims[nb])
        x, gx, cdown = b1.z, b1.gz, feats[nb]
        self.segs = []
        for u in range(num_pool):
            d = num_pool - 1 - u
            cs = feats[d]
            up = _Blk(kind="up", cin=cdown, cout=cs, strides=pools[-(u + 1)], x=x, gx=gx, y=Act(self.cat[u], 0, cs),
                      gy=Act(self.gcat[u], 0, cs), w=by[f"tu.{u}.weight"])
            self.order.append(up)
            b0 = block(f"conv_blocks_localization.{u}.0.blocks.0", 2 * cs, cs, one, kernels[-(u + 1)], Act(self.cat[u], 0, 2 * cs),
                       Act(self.gcat[u], 0, 2 * cs), False, None, None, dims[d])
            b1 = block(f"conv_blocks_localization.{u}.1.blocks.0", cs, cs, one, kernels[-(u + 1)], b0.z, b0.gz, False, None, None, dims[d])
            seg = _Blk(kind="seg", cin=cs, x=b1.z, gx=b1.gz, gx_has_prior=(u < num_pool - 1), w=by[f"seg_outputs.{u}.weight"])
            self.order.append(seg)
            self.segs.append(seg)
            x, gx, cdown = b1.z, b1.gz, cs
        self.ws = torch.zeros(2 * N * max(2 * f for f in feats), dtype=torch.float64, device=dev)
        self.unused_heads: List[str] = []

    def pview(self, slot, arena=None):
        a = self.arena.theta if arena is None else arena
        return a[slot.offset:slot.offset + slot.numel].view(slot.shape)

    @staticmethod
    def _p(act: Act):
        return _Ptr32(act.buf, act.off)

    def forward(self, x: torch.Tensor, seg_weights: Optional[List[torch.Tensor]] = None, body: bool = True):
        N = self.N
        assert tuple(x.shape) == (N, self.in_channels) + self.patch
        if body:
            if self.in_channels == 1:        # (N, 1, D, H, W) fp32 == NDHWC with one channel
                self.image = x.contiguous()
            else:
                if self.image is None or self.image.shape[0] != N:
                    self.image = torch.empty((N,) + self.patch + (self.in_channels,), device=self.device)
                nat.call("lnn_f32_image_to_cl", x.contiguous(), self.image, N, self.in_channels, x[0, 0].numel(), self.in_channels)
        logits = [torch.empty((N, self.K) + seg.x.dims, device=self.device) for seg in self.segs]
        sw = None if seg_weights is None else [w.contiguous() for w in seg_weights]
        u = 0
        for it in self.order:
            if it.kind == "conv":
                if not body:
                    continue
                D, H, W = it.in_dims
                xin = self.image if it.x is None else self._p(it.x)
                ldx = self.in_channels if it.x is None else it.x.ld
                nat.call("lnn_f32_conv3d_fwd_g", xin, ldx, self.pview(it.w), self.pview(it.b), it.y, it.cout, N, D, H, W, it.cin,
                         it.cout, *it.kernel, *it.strides)
                nat.call("lnn_f32_instnorm_lrelu_fwd", it.y, it.cout, self._p(it.z), it.z.ld, N, it.z.V, it.cout, IN_EPS, it.mean,
                         it.rstd, self.pview(it.gamma), self.pview(it.beta), LRELU_SLOPE)
            elif it.kind == "up":
                if not body:
                    continue
                D, H, W = it.x.dims
                nat.call("lnn_f32_convT3d_fwd_g", self._p(it.x), it.x.ld, self.pview(it.w), self._p(it.y), it.y.ld, N, D, H, W,
                         it.cin, it.cout, *it.strides)
            else:
                w = self.pview(it.w) if sw is None else sw[u]
                nat.call("lnn_f32_seg1x1_fwd", self._p(it.x), it.x.ld, w, logits[u], N, it.x.V, it.cin, self.K)
                u += 1
        return logits

    def conv_outputs(self, logits):
        """Same contract as UNetEngine.conv_outputs (fp32 views)."""
        from collections import OrderedDict
        out, u = OrderedDict(), 0
        for it in self.order:
            if it.kind == "conv":
                out[it.prefix + ".conv"] = it.y.permute(0, 4, 1, 2, 3)
            elif it.kind == "up":
                out[f"tu.{u}"] = it.y.tensor().permute(0, 4, 1, 2, 3)
            else:
                out[f"seg_outputs.{u}"] = logits[u]
                u += 1
        return out

    def backward(self, dlogits, skip_body: bool = False, progress=None):
        N = self.N
        grad = self.arena.grad
        self.unused_heads = [seg.w.name for seg, dl in zip(self.segs, dlogits) if dl is None]
        dls = [None if dl is None else dl.contiguous() for dl in dlogits]
        seg_u = len(self.segs)
        for it in reversed(self.order):
            if it.kind == "seg":
                seg_u -= 1
                dl = dls[seg_u]
                if dl is None:
                    if not it.gx_has_prior:
                        it.gx.buf.zero_()
                    continue
                nat.call("lnn_f32_seg1x1_bwd", self._p(it.x), it.x.ld, self.pview(it.w), dl, self._p(it.gx), it.gx.ld,
                         self.pview(it.w, grad), N, it.x.V, it.cin, self.K, 1 if it.gx_has_prior else 0)
            elif skip_body:
                continue
            elif it.kind == "conv":
                nat.call("lnn_f32_instnorm_lrelu_bwd", it.y, it.cout, self._p(it.gz), it.gz.ld, N, it.z.V, it.cout, it.mean, it.rstd,
                         self.pview(it.gamma), self.pview(it.beta), LRELU_SLOPE, self.pview(it.gamma, grad),
                         self.pview(it.beta, grad), self.ws)
                D, H, W = it.in_dims
                xin = self.image if it.x is None else self._p(it.x)
                ldx = self.in_channels if it.x is None else it.x.ld
                nat.call("lnn_f32_conv3d_wgrad_g", xin, ldx, it.y, it.cout, self.pview(it.w, grad), N, D, H, W, it.cin, it.cout,
                         *it.kernel, *it.strides)
                if it.x is not None and it.gx is not None:
                    nat.call("lnn_f32_conv3d_dgrad_g", it.y, it.cout, self.pview(it.w), self._p(it.gx), it.gx.ld, N, D, H, W, it.cin,
                             it.cout, *it.kernel, *it.strides, 1 if it.gx_accumulate else 0)
            else:
                D, H, W = it.x.dims
                nat.call("lnn_f32_convT3d_wgrad_g", self._p(it.x), it.x.ld, self._p(it.gy), it.gy.ld, self.pview(it.w, grad), N,
                         D, H, W, it.cin, it.cout, *it.strides)
                nat.call("lnn_f32_convT3d_dgrad_g", self._p(it.gy), it.gy.ld, self.pview(it.w), self._p(it.gx), it.gx.ld, N, D, H,
                         W, it.cin, it.cout, *it.strides, 0)
        if progress is not None:
            progress(0)

    def flops_per_patch(self):
        mac = first = 0
        for it in self.order:
            if it.kind == "conv":
                m = it.z.V * it.cin * it.cout * it.kernel[0] * it.kernel[1] * it.kernel[2]
                mac += m
                if it.x is None:
                    first = m
            elif it.kind == "up":
                mac += it.x.V * it.cin * it.cout * it.strides[0] * it.strides[1] * it.strides[2]
            else:
                mac += it.x.V * it.cin * self.K
        return 6 * mac - 2 * first, mac


class _Ptr32:
    def __init__(self, t, off):
        self.t, self.off = t, off

    def data_ptr(self):
        return self.t.data_ptr() + 4 * self.off
