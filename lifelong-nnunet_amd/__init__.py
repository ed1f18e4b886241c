"""lifelong-nnunet_amd: MI355X-native 3-D U-Net training step + EWC / LwF regularisers.

Drop-in for the hot path of MECLabTUDA/Lifelong-nnUNet's ``nnUNetTrainer*`` plugins: Python host code
on PyTorch-ROCm (device memory, streams, torch.distributed) calling hand-written gfx950 HIP kernels
through the C-ABI library declared in ``include/lnn_hip.h``.
"""
__version__ = "0.1.0"

# plugin surface: same class names / module layout as nnunet_ext/training/network_training/<ext>/ (run_training.py:18-28)
TRAINER_MAP = {
    "multihead": "lifelong_nnunet_amd.training.network_training.multihead.nnUNetTrainerMultiHead:nnUNetTrainerMultiHead",
    "sequential": "lifelong_nnunet_amd.training.network_training.sequential.nnUNetTrainerSequential:nnUNetTrainerSequential",
    "ewc": "lifelong_nnunet_amd.training.network_training.ewc.nnUNetTrainerEWC:nnUNetTrainerEWC",
    "rw": "lifelong_nnunet_amd.training.network_training.rw.nnUNetTrainerRW:nnUNetTrainerRW",
    "mib": "lifelong_nnunet_amd.training.network_training.mib.nnUNetTrainerMiB:nnUNetTrainerMiB",
    "plop": "lifelong_nnunet_amd.training.network_training.plop.nnUNetTrainerPLOP:nnUNetTrainerPLOP",
    "pod": "lifelong_nnunet_amd.training.network_training.pod.nnUNetTrainerPOD:nnUNetTrainerPOD",
    "lwf": "lifelong_nnunet_amd.training.network_training.lwf.nnUNetTrainerLWF:nnUNetTrainerLWF",
    "rehearsal": "lifelong_nnunet_amd.training.network_training.rehearsal.nnUNetTrainerRehearsal:nnUNetTrainerRehearsal",
    "rehearsal_ewc": "lifelong_nnunet_amd.training.network_training.rehearsal_ewc.nnUNetTrainerRehearsalEWC:nnUNetTrainerRehearsalEWC",
}


def get_trainer_class(extension):
    import importlib
    mod, cls = TRAINER_MAP[extension].split(":")
    return getattr(importlib.import_module(mod), cls)
