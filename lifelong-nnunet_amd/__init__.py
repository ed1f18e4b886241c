"""lifelong-nnunet_amd: MI355X-native 3-D U-Net training step + EWC / LwF regularisers.

Drop-in for the hot path of MECLabTUDA/Lifelong-nnUNet's ``nnUNetTrainer*`` plugins: Python host code
on PyTorch-ROCm (device memory, streams, torch.distributed) calling hand-written gfx950 HIP kernels
through the C-ABI library declared in ``include/lnn_hip.h``.
"""
__version__ = "0.1.0"
