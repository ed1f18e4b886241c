"""Generates tests/golden/forward_wiring_reference.npz: the U-Net forward WIRING executed by the reference's own lines.

Upstream ``Generic_UNet.forward`` is not in the reference tree, but the reference's ``Generic_ViT_UNet.forward``
(nnunet_ext/network_architecture/generic_ViT_UNet.py:216-284) carries it ("Copied from original implementation": the encoder loop that
collects the skips, the decoder loop ``tu -> cat((x, skip), 1) -> conv_blocks_localization -> final_nonlin(seg_outputs)``, the
deep-supervision tuple ``[last] + upscaled(reversed(rest))``) around its transformer.  Here that function is imported from
/root/reference and called UNBOUND on the oracle network, with the transformer stood in by the identity (``version='V1'``, a
``prepare`` entry that hands the bottleneck tensor through, ``ViT = identity``; ``final_nonlin`` and ``upscale_logits_ops`` are the
identities nnU-Net's trainer configures).  What the fixture pins: the ORDER of everything in the oracle's ``forward`` -- skip indexing,
concatenation order, which decoder level feeds which segmentation layer, the order of the returned tuple -- not the insides of the conv
blocks (upstream ``ConvDropoutNormNonlin``: restated, still unpinned).

    python -m oracle.make_goldens_forward        (in the build container; /root/reference is not on the GPU box)

Only DATA is written (npz): no reference source or bytecode is copied."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import ref_shim
from .make_goldens import OUT
from .unet import OracleGenericUNet

CTOR = (1, 4, 3, 3)


def main():
    ref_shim.install()
    from nnunet_ext.network_architecture.generic_ViT_UNet import Generic_ViT_UNet
    torch.manual_seed(31)
    net = OracleGenericUNet(*CTOR)
    net.eval()
    with torch.no_grad():                       # distinguishable levels: every parameter tensor gets its own offset
        for i, p in enumerate(net.parameters()):
            p.add_(0.01 * ((i % 5) - 2))
    x = torch.randn(2, 1, 8, 16, 8)
    stand_ins = dict(convolutional_pooling=True, version="V1", split_gpu=False, prepare={"V1": "_lnn_bottleneck_through"},
                     _lnn_bottleneck_through=lambda skips, last: last, ViT=lambda v: v, final_nonlin=lambda t: t,
                     upscale_logits_ops=[(lambda t: t)] * (CTOR[3] - 1))
    for k, v in stand_ins.items():
        object.__setattr__(net, k, v)           # plain attributes: the module tree of the oracle network is not touched
    with torch.no_grad():
        net.do_ds = True
        ref_ds = Generic_ViT_UNet.forward(net, x)
        net.do_ds = False
        ref_single = Generic_ViT_UNet.forward(net, x)
        net.do_ds = True
        own = net(x)
    assert isinstance(ref_ds, tuple) and len(ref_ds) == CTOR[3]
    arrays = {"x": x.numpy(), "ctor": np.array(CTOR), "ref_single": ref_single.numpy()}
    for i, t in enumerate(ref_ds):
        arrays[f"ref_ds_{i}"] = t.numpy()
    for k, v in net.state_dict().items():
        arrays["sd::" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "forward_wiring_reference.npz"), **arrays)
    print("levels:", [tuple(t.shape) for t in ref_ds], "| oracle forward equal to the reference's lines:",
          all(torch.equal(a, b) for a, b in zip(ref_ds, own)))


if __name__ == "__main__":
    main()
