"""``nnUNetTrainerSequential`` -- the "plain sequential trainer" of BASELINE config 2.

Mirror of nnunet_ext/training/network_training/sequential/nnUNetTrainerSequential.py:19-82: a MultiHead
trainer with ``transfer_heads=True`` forced (:32; new head initialised from the last head) whose U-Net
children are re-registered encoder -> decoder -> head (:65, MH.py:1391-1408) -- parameter ORDER only.
"""
from ..multihead.nnUNetTrainerMultiHead import nnUNetTrainerMultiHead

HYPERPARAMS = {}


class nnUNetTrainerSequential(nnUNetTrainerMultiHead):
    def __init__(self, split, task, *args, **kwargs):
        kwargs["transfer_heads"] = True
        kwargs.setdefault("extension", "sequential")
        super().__init__(split, task, *args, **kwargs)

    def initialize_network(self):
        super().initialize_network()
        self.reorder_UNet_components()
