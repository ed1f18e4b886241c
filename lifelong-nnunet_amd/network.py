"""``Generic_UNet`` drop-in whose forward/backward run on the hand-written gfx950 kernels.

Mirrors the network contract the reference's trainers rely on:
  * ``net(x)`` returns a tuple of ``num_pool`` logits tensors, full resolution first
    (generic_ViT_UNet.py:282-284), or only the full-resolution one when ``do_ds`` is False;
  * attributes ``do_ds`` / ``inference_apply_nonlin`` toggled by callers (lwf/nnUNetTrainerLWF.py:322,355);
  * ``named_parameters()`` names, shapes AND order equal upstream's (module registration order
    conv_blocks_localization, conv_blocks_context, td, tu, seg_outputs --
    test_MultiHead_Module.py:283,345,417,422,427): Fisher / theta* dictionaries are keyed by them
    (ewc/nnUNetTrainerEWC.py:298-304), MultiHead_Module freezes and splits on them.
Every parameter is a view into one flat fp32 arena, every ``.grad`` a view into a flat gradient arena;
backward accumulates into that arena (``optimizer.zero_grad()`` = one memset).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
from torch import nn

from .engine import ParamArena, UNetEngine, param_slots, unet_geometry
from .engine_f32 import UNetEngineF32


class _ParamHolder(nn.Module):
    """conv / instnorm leaf: holds ``weight`` (and ``bias``) parameters that view into the arena."""

    def __init__(self, weight, bias=None):
        super().__init__()
        self.weight = weight
        if bias is not None:
            self.bias = bias


class ConvDropoutNormNonlin(nn.Module):
    def __init__(self, conv, instnorm):
        super().__init__()
        self.conv = conv
        self.instnorm = instnorm
        self.lrelu = nn.LeakyReLU(1e-2, inplace=True)   # parameter-free, kept for tree fidelity


class StackedConvLayers(nn.Module):
    def __init__(self, blocks, cin, cout):
        super().__init__()
        self.input_channels, self.output_channels = cin, cout
        self.blocks = nn.Sequential(*blocks)


class _UNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, net, *params):
        eng = net.engine_for(x)
        logits = eng.forward(x)
        ctx.net, ctx.eng = net, eng
        ctx.set_materialize_grads(False)
        return tuple(logits[::-1])                      # full resolution first

    @staticmethod
    def backward(ctx, *gouts):
        net, eng = ctx.net, ctx.eng
        net.bind_grads()
        body_needs_grad = any(p.requires_grad for n, p in net._named if not n.startswith("seg_outputs."))
        eng.backward(list(gouts[::-1]), skip_body=not body_needs_grad, progress=getattr(net, "on_grad_progress", None))
        net.params_without_grad = set(eng.unused_heads)
        return (None, None) + (None,) * len(net._named)


class Generic_UNet(nn.Module):
    MAX_NUM_FILTERS_3D = 320

    def __init__(self, input_channels, base_num_features, num_classes, num_pool, num_conv_per_stage=2,
                 deep_supervision=True, device="cuda", pool_op_kernel_sizes=None, conv_kernel_sizes=None, **_ignored):
        """``pool_op_kernel_sizes`` / ``conv_kernel_sizes``: the plans' per-level poolings (strides, 1 or 2 per axis) and conv
        kernel extents (1 or 3 per axis) the reference hands to its network class (nnViTUNetTrainer.py:117-122 /
        nnUNetTrainerMultiHead.py:348-369); None = 2x2x2 / 3x3x3 everywhere (upstream's defaults)."""
        super().__init__()
        self.input_channels, self.base_num_features = input_channels, base_num_features
        self.num_classes, self.num_pool = num_classes, num_pool
        self.pool_op_kernel_sizes, self.conv_kernel_sizes, _ = unet_geometry(num_pool, None, pool_op_kernel_sizes, conv_kernel_sizes)
        self._deep_supervision = deep_supervision
        self.do_ds = True
        self.inference_apply_nonlin = lambda x: torch.softmax(x, 1)
        self.convolutional_pooling = self.convolutional_upsampling = True
        self.device_ = torch.device(device)
        self.storage = "fp16"      # "fp32": the reference's fp16=False branch (MH.py:632-641) on the direct fp32 kernels

        slots = param_slots(input_channels, base_num_features, num_classes, num_pool, self.MAX_NUM_FILTERS_3D,
                            self.pool_op_kernel_sizes, self.conv_kernel_sizes)
        self.arena = ParamArena(slots, self.device_)
        self._engines: Dict[Tuple, UNetEngine] = {}
        self.params_without_grad = set()

        P: Dict[str, nn.Parameter] = {}
        for s in slots:
            p = nn.Parameter(self.arena.view(s))
            p._lnn_net, p._lnn_slot = self, s             # lets the losses find the arena from (name, param)
            P[s.name] = p

        def block(prefix, cin, cout):
            return ConvDropoutNormNonlin(_ParamHolder(P[prefix + ".conv.weight"], P[prefix + ".conv.bias"]),
                                         _ParamHolder(P[prefix + ".instnorm.weight"], P[prefix + ".instnorm.bias"]))

        feats = [min(base_num_features * 2 ** d, self.MAX_NUM_FILTERS_3D) for d in range(num_pool + 1)]
        # registration order as upstream: localization, context, td, tu, seg_outputs
        self.conv_blocks_localization = nn.ModuleList()
        self.conv_blocks_context = nn.ModuleList()
        self.td = nn.ModuleList()
        self.tu = nn.ModuleList()
        self.seg_outputs = nn.ModuleList()
        cin = input_channels
        for d in range(num_pool):
            pre = f"conv_blocks_context.{d}.blocks"
            self.conv_blocks_context.append(StackedConvLayers(
                [block(pre + ".0", cin, feats[d]), block(pre + ".1", feats[d], feats[d])], cin, feats[d]))
            cin = feats[d]
        nb = num_pool
        self.conv_blocks_context.append(nn.Sequential(
            StackedConvLayers([block(f"conv_blocks_context.{nb}.0.blocks.0", cin, feats[nb])], cin, feats[nb]),
            StackedConvLayers([block(f"conv_blocks_context.{nb}.1.blocks.0", feats[nb], feats[nb])], feats[nb], feats[nb])))
        for u in range(num_pool):
            cs = feats[num_pool - 1 - u]
            self.tu.append(_ParamHolder(P[f"tu.{u}.weight"]))
            self.conv_blocks_localization.append(nn.Sequential(
                StackedConvLayers([block(f"conv_blocks_localization.{u}.0.blocks.0", 2 * cs, cs)], 2 * cs, cs),
                StackedConvLayers([block(f"conv_blocks_localization.{u}.1.blocks.0", cs, cs)], cs, cs)))
            self.seg_outputs.append(_ParamHolder(P[f"seg_outputs.{u}.weight"]))
        self._named = list(self.named_parameters())
        assert len(self._named) == len(slots)
        self.bind_grads()
        self.reset_parameters()

    # ------------------------------------------------------------------------------------------ params
    def reset_parameters(self):
        """upstream InitWeights_He(1e-2) (arg at nnViTUNetTrainer.py:121): kaiming_normal(a=0.01) for conv and
        transposed-conv weights (fan_in = weight.size(1) * receptive field), zero biases, IN gamma=1 beta=0."""
        with torch.no_grad():
            for name, p in self._named:
                if name.endswith("conv.weight") or name.startswith("tu.") or name.startswith("seg_outputs."):
                    w = torch.empty(p.shape)
                    nn.init.kaiming_normal_(w, a=1e-2)
                    p.copy_(w)
                elif name.endswith("instnorm.weight"):
                    p.fill_(1.0)
                else:
                    p.zero_()
        self.mark_params_changed()

    def mark_params_changed(self):
        self.arena.version += 1

    def bind_grads(self):
        for name, p in self._named:
            if p.grad is None:
                p.grad = self.arena.view(p._lnn_slot, "grad")

    def load_state_dict(self, state_dict, strict=True):
        r = super().load_state_dict(state_dict, strict)
        self.mark_params_changed()
        return r

    def _apply(self, fn, *a, **k):
        # parameters live in the arena on the construction device; .to()/.cuda() must not re-home them
        return self

    # ------------------------------------------------------------------------------------------ engine
    def engine_for(self, x) -> UNetEngine:
        key = (x.shape[0],) + tuple(x.shape[2:]) + ((self.storage,) if self.storage != "fp16" else ())
        eng = self._engines.get(key)
        if eng is None:
            cls = UNetEngineF32 if self.storage == "fp32" else UNetEngine
            eng = cls(self.arena, self.input_channels, self.base_num_features, self.num_classes, self.num_pool,
                      tuple(x.shape[2:]), x.shape[0], self.device_, self.MAX_NUM_FILTERS_3D,
                      pool_op_kernel_sizes=self.pool_op_kernel_sizes, conv_kernel_sizes=self.conv_kernel_sizes)
            if getattr(self, "deterministic_wgrad", False) and hasattr(eng, "deterministic_wgrad"):
                eng.deterministic_wgrad = True        # ordered reduction instead of fp32 atomics in every weight gradient
            self._engines[key] = eng
        return eng

    def forward(self, x):
        x = x.to(self.device_, torch.float32)
        if torch.is_grad_enabled() and any(p.requires_grad for _, p in self._named):
            outs = _UNetFunction.apply(x, self, *[p for _, p in self._named])
        else:
            outs = tuple(self.engine_for(x).forward(x)[::-1])
        if self._deep_supervision and self.do_ds:
            return outs
        return outs[0]

    def forward_heads(self, x, head_weights):
        """Full-resolution logits of several heads on ONE body pass (the reference runs a complete eval
        forward per head, lwf/nnUNetTrainerLWF.py:315-346; InstanceNorm has no running statistics, so the
        body activations are identical in eval and train mode).  ``head_weights``: list of lists of the
        ``num_pool`` seg_outputs weights per head."""
        x = x.to(self.device_, torch.float32)
        eng = self.engine_for(x)
        outs = []
        with torch.no_grad():
            for i, hw in enumerate(head_weights):
                outs.append(eng.forward(x, seg_weights=hw, body=(i == 0))[-1])
        return outs
