"""``nnUNetTrainerMiB`` -- Modeling the Background (unbiased knowledge distillation against the previous model).

Mirror of nnunet_ext/training/network_training/mib/nnUNetTrainerMiB.py: constructor :23-58 (``HYPERPARAMS`` :21),
``initialize`` :60-75 (``loss_orig`` = the deep-supervised Dice+CE, ``loss_mib`` = MultipleOutputLossMiB),
``run_training`` :91-103 (snapshot of the network as ``network_old`` when a NEW task starts), ``run_iteration``
:105-182 (first task and validation: the original loss; otherwise forward of the current and of the old model on the
same batch and ``loss_mib(output, output_o, target)``).

Both forwards are the HIP engine (the old model is a second ``Generic_UNet`` with its own parameter arena and weight
panels, evaluated without autograd); CE and the distillation term are the fused ``lnn_target_ce_{fwd,bwd}`` kernels,
one launch per deep-supervision level and direction.
"""
import torch

from ....losses import MultipleOutputLossMiB as MiBLoss
from ..multihead.nnUNetTrainerMultiHead import nnUNetTrainerMultiHead

HYPERPARAMS = {'mib_alpha': float, 'mib_lkd': float}


class nnUNetTrainerMiB(nnUNetTrainerMultiHead):
    def __init__(self, split, task, *args, mib_alpha=1., mib_lkd=10, **kwargs):
        kwargs.setdefault("extension", "mib")
        super().__init__(split, task, *args, **kwargs)
        self.alpha, self.lkd = mib_alpha, mib_lkd
        self.network_old = None
        self._x_o = None

    def initialize(self, training=True, force_load_plans=False, num_epochs=500, prev_trainer_path=None,
                   call_for_eval=False):
        super().initialize(training, force_load_plans, num_epochs, prev_trainer_path, call_for_eval)
        self.loss_orig = self.loss
        self.loss_mib = MiBLoss(self.alpha, self.lkd, self.ds_loss_weights)

    def run_training(self, task, output_folder=None, build_folder=True):
        if not self.was_initialized:
            self.initialize(True, num_epochs=self.max_num_epochs)
        if str(task) not in self.mh_network.heads:
            self.network_old = self.frozen_copy_of_network()         # copy.deepcopy(self.network), MiB.py:96
        return super().run_training(task, output_folder, build_folder)

    def on_forward_done(self, data, output, do_backprop):
        if self._use_mib:
            with torch.no_grad():
                self._x_o = tuple(o.detach() for o in self.network_old(data))

    def run_iteration(self, data_generator, do_backprop=True, run_online_evaluation=False, detach=True, no_loss=False):
        first_task = str(self.task) in self.mh_network.heads and len(self.mh_network.heads) == 1
        self._use_mib = not (first_task or run_online_evaluation) and self.network_old is not None
        if self._use_mib:
            self.loss = lambda output, target: self.loss_mib(output, self._x_o, target)
        else:
            self.loss = self.loss_orig
        try:
            return super().run_iteration(data_generator, do_backprop, run_online_evaluation, detach, no_loss)
        finally:
            self._x_o = None
