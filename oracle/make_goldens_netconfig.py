"""Generates tests/golden/network_config_reference.json: the configuration the REFERENCE hands to its network class.

``nnViTUNetTrainer.initialize_network`` (nnunet_ext/training/network_training/nnViTUNetTrainer.py:97-122; its first half is upstream
nnUNetTrainerV2's, "copied from original implementation") is imported from /root/reference and EXECUTED on a stand-in trainer object
with the network class and the weight initialiser replaced by recorders.  What comes out is the argument list of the network
constructor as the reference builds it: the conv / norm / dropout / nonlinearity classes and their keyword arguments (InstanceNorm
eps 1e-5 affine, Dropout p = 0, LeakyReLU slope 1e-2), deep supervision on, no dropout in the localisation path, identity final
nonlinearity, He initialisation with a = 1e-2, no logits upscaling, convolutional pooling AND convolutional upsampling.
tests/test_host_logic.py::test_network_configuration_matches_the_reference holds the oracle network and the product's constants to it.
A second run records a Prostate-shaped plan (two input channels, anisotropic poolings / kernels): the per-level lists reach the
network class unchanged (``prostate_shaped`` key; test_anisotropic_plan_builds_the_reference_configuration).

    python -m oracle.make_goldens_netconfig        (in the build container; /root/reference is not on the GPU box)

Only DATA is written (json): no reference source or bytecode is copied."""
from __future__ import annotations

import json
import os
import types

import numpy as np
import torch

from . import ref_shim
from .make_goldens import OUT


class _Recorder:
    calls = []

    def __init__(self, *a, **k):
        type(self).calls.append((a, k))
        self.ViT = types.SimpleNamespace(use_task=lambda t: None)


class _HeRecorder:
    def __init__(self, neg_slope=1e-2):
        self.neg_slope = neg_slope


def _record(mod, me):
    _Recorder.calls.clear()
    with ref_shim.cuda_as_cpu():
        mod.nnViTUNetTrainer.initialize_network(me)
    (a, k), = _Recorder.calls
    name = lambda c: c.__module__ + "." + c.__name__
    probe = torch.tensor([-2.0, 0.5])
    return {"input_channels": a[0], "base_num_features": a[1], "num_classes": a[2], "num_pool": a[3], "patch_size": list(a[4]),
            "num_conv_per_stage": a[5], "feat_map_mul_on_downscale": a[6], "conv_op": name(a[7]), "norm_op": name(a[8]),
            "norm_op_kwargs": a[9], "dropout_op": name(a[10]), "dropout_op_kwargs": a[11], "nonlin": name(a[12]), "nonlin_kwargs": a[13],
            "deep_supervision": a[14], "dropout_in_localization": a[15], "final_nonlin_of_probe": a[16](probe).tolist(),
            "he_init_neg_slope": a[17].neg_slope, "pool_op_kernel_sizes": a[18], "conv_kernel_sizes": a[19], "upscale_logits": a[20],
            "convolutional_pooling": a[21], "convolutional_upsampling": a[22], "n_positional": len(a), "keywords": sorted(k)}


def main():
    ref_shim.install()
    import nnunet_ext.training.network_training.nnViTUNetTrainer as mod
    mod.Generic_ViT_UNet, mod.InitWeights_He = _Recorder, _HeRecorder
    common = dict(threeD=True, conv_per_stage=2, version="V1", vit_type="base", split_gpu=False, ViT_task_specific_ln=False,
                  first_task_name="t", LSA=False, SPT=False)
    pool = [[2, 2, 2]] * 5
    convk = [[3, 3, 3]] * 6
    me = types.SimpleNamespace(num_input_channels=1, base_num_features=32, num_classes=3, net_num_pool_op_kernel_sizes=pool,
                               patch_size=np.array([160, 192, 160]), net_conv_kernel_sizes=convk, **common)
    cfg = _record(mod, me)
    # a plan shaped like Task005_Prostate's 3d_fullres (two modalities, thin slabs: the first poolings and kernels leave z alone;
    # documentation/setting_up_paths.md:24-28, README.md:73), at toy size: what reaches the network class for such a plan
    me = types.SimpleNamespace(num_input_channels=2, base_num_features=8, num_classes=3,
                               net_num_pool_op_kernel_sizes=[[1, 2, 2], [1, 2, 2], [2, 2, 2]], patch_size=np.array([8, 32, 32]),
                               net_conv_kernel_sizes=[[1, 3, 3], [1, 3, 3], [3, 3, 3], [3, 3, 3]], **common)
    cfg["prostate_shaped"] = _record(mod, me)
    json.dump(cfg, open(os.path.join(OUT, "network_config_reference.json"), "w"), indent=1)
    print(json.dumps(cfg)[:900])


if __name__ == "__main__":
    main()
