"""Oracle restatement of ``Generic_UNet`` as nnUNetTrainerV2 configures it (CPU, fp32, plain torch).

Follows (reference paths relative to /root/reference):
  * forward:      nnunet_ext/network_architecture/generic_ViT_UNet.py:222-230,261-264,280-286
  * ctor args:    nnunet_ext/training/network_training/nnViTUNetTrainer.py:101-125
  * module tree / parameter names: test/network_architecture/test_MultiHead_Module.py:281-433
  * upstream nnunet @77bc485 ``Generic_UNet.__init__`` (recalled)

TEST INFRASTRUCTURE ONLY -- never imported by the product package.
"""
from __future__ import annotations

import torch
from torch import nn


class ConvDropoutNormNonlin(nn.Module):
    # attribute names conv / instnorm / lrelu as in test_MultiHead_Module.py:287-291; dropout p=0 is omitted upstream
    def __init__(self, cin, cout, stride, kernel=(3, 3, 3)):
        super().__init__()
        # upstream: conv_kwargs['kernel_size'] = conv_kernel_sizes[level], 'padding' = [1 if k == 3 else 0 for k in kernel]
        self.conv = nn.Conv3d(cin, cout, tuple(kernel), stride, tuple(1 if k == 3 else 0 for k in kernel), bias=True)
        self.instnorm = nn.InstanceNorm3d(cout, eps=1e-5, affine=True)
        self.lrelu = nn.LeakyReLU(1e-2, inplace=True)

    def forward(self, x):
        return self.lrelu(self.instnorm(self.conv(x)))


class StackedConvLayers(nn.Module):
    def __init__(self, cin, cout, num_convs, first_stride=1, kernel=(3, 3, 3)):
        super().__init__()
        self.input_channels, self.output_channels = cin, cout
        self.blocks = nn.Sequential(
            *([ConvDropoutNormNonlin(cin, cout, first_stride, kernel)]
              + [ConvDropoutNormNonlin(cout, cout, 1, kernel) for _ in range(num_convs - 1)]))

    def forward(self, x):
        return self.blocks(x)


def init_weights_he(module, neg_slope=1e-2):
    # upstream InitWeights_He(1e-2) (arg at nnViTUNetTrainer.py:121)
    if isinstance(module, (nn.Conv3d, nn.ConvTranspose3d)):
        module.weight = nn.init.kaiming_normal_(module.weight, a=neg_slope)
        if module.bias is not None:
            module.bias = nn.init.constant_(module.bias, 0)


class OracleGenericUNet(nn.Module):
    """3-D Generic_UNet: conv pooling, transposed-conv upsampling, deep supervision, no logits upscaling."""
    MAX_FEATURES_3D = 320

    def __init__(self, in_channels, base_features, num_classes, num_pool, conv_per_stage=2, pool_op_kernel_sizes=None,
                 conv_kernel_sizes=None):
        """``pool_op_kernel_sizes`` / ``conv_kernel_sizes``: as the reference hands them over (nnViTUNetTrainer.py:122); the way
        upstream's constructor (nnunet @77bc485, absent from the reference tree: recalled, parity unpinned) uses them: encoder
        stage d = kernel d, first block strided by pooling d - 1; bottleneck = kernel num_pool, pooling -1; decoder stage u =
        ConvTranspose3d(kernel = stride = pooling -(u + 1)) and conv kernel -(u + 1)."""
        super().__init__()
        pools = [tuple(q) for q in (pool_op_kernel_sizes if pool_op_kernel_sizes is not None else [(2, 2, 2)] * num_pool)]
        kernels = [tuple(q) for q in (conv_kernel_sizes if conv_kernel_sizes is not None else [(3, 3, 3)] * (num_pool + 1))]
        assert len(pools) == num_pool and len(kernels) == num_pool + 1
        self.num_classes = num_classes
        self.do_ds = True
        self._deep_supervision = True
        self.inference_apply_nonlin = lambda x: torch.softmax(x, 1)

        # registration order: localization, context, td, tu, seg_outputs (test_MultiHead_Module.py:283,345,417,422,427)
        self.conv_blocks_localization = nn.ModuleList()
        self.conv_blocks_context = nn.ModuleList()
        self.td = nn.ModuleList()
        self.tu = nn.ModuleList()
        self.seg_outputs = nn.ModuleList()

        cin, cout = in_channels, base_features
        for d in range(num_pool):
            self.conv_blocks_context.append(StackedConvLayers(cin, cout, conv_per_stage, pools[d - 1] if d > 0 else 1, kernels[d]))
            cin = cout
            cout = min(cout * 2, self.MAX_FEATURES_3D)
        # bottleneck (test_MultiHead_Module.py:394-415): Sequential(Stacked(strided, n-1 convs), Stacked(1 conv))
        final = cout  # convolutional_upsampling=True -> final_num_features = output_features
        self.conv_blocks_context.append(nn.Sequential(
            StackedConvLayers(cin, cout, conv_per_stage - 1, pools[-1], kernels[num_pool]),
            StackedConvLayers(cout, final, 1, 1, kernels[num_pool])))

        for u in range(num_pool):
            from_down = final
            from_skip = self.conv_blocks_context[-(2 + u)].output_channels
            final = from_skip
            self.tu.append(nn.ConvTranspose3d(from_down, from_skip, pools[-(u + 1)], pools[-(u + 1)], bias=False))
            self.conv_blocks_localization.append(nn.Sequential(
                StackedConvLayers(2 * from_skip, from_skip, conv_per_stage - 1, 1, kernels[-(u + 1)]),
                StackedConvLayers(from_skip, final, 1, 1, kernels[-(u + 1)])))
        for u in range(num_pool):
            self.seg_outputs.append(nn.Conv3d(self.conv_blocks_localization[u][-1].output_channels,
                                              num_classes, 1, 1, 0, bias=False))
        self.apply(init_weights_he)

    def forward(self, x):
        skips, seg_outputs = [], []
        for d in range(len(self.conv_blocks_context) - 1):
            x = self.conv_blocks_context[d](x)
            skips.append(x)
        x = self.conv_blocks_context[-1](x)
        for u in range(len(self.tu)):
            x = self.tu[u](x)
            x = torch.cat((x, skips[-(u + 1)]), dim=1)
            x = self.conv_blocks_localization[u](x)
            seg_outputs.append(self.seg_outputs[u](x))
        if self._deep_supervision and self.do_ds:
            return tuple([seg_outputs[-1]] + list(seg_outputs[:-1][::-1]))
        return seg_outputs[-1]
