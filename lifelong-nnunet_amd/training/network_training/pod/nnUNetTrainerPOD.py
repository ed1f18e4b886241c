"""``nnUNetTrainerPOD`` -- local POD distillation only (PLOP without the pseudo labels).

Mirror of nnunet_ext/training/network_training/pod/nnUNetTrainerPOD.py: constructor :21-37, ``initialize`` :39-51
(``loss_plop`` = MultipleOutputLossPOD around the Dice+CE base loss), ``run_training`` :58-82 (snapshot of the network,
no threshold extraction), ``run_iteration`` :84-96 (the PLOP iteration with ``pod=True``).  The third-task hook aliasing of
the PLOP trainer applies here as well (tests/golden/plop_reference.json:pod_flow.pods_taskC is all zeros).
"""
from ....losses import DC_and_CE_loss, MultipleOutputLossPOD as PODLoss
from ..multihead.nnUNetTrainerMultiHead import nnUNetTrainerMultiHead
from ..plop.nnUNetTrainerPLOP import nnUNetTrainerPLOP

HYPERPARAMS = {'pod_lambda': float, 'pod_scales': int}


class nnUNetTrainerPOD(nnUNetTrainerPLOP):
    def __init__(self, split, task, *args, **kwargs):
        kwargs.setdefault("extension", "pod")
        super().__init__(split, task, *args, **kwargs)
        del self.thresholds, self.max_entropy           # POD.py:37

    def initialize(self, training=True, force_load_plans=False, num_epochs=500, prev_trainer_path=None,
                   call_for_eval=False):
        super().initialize(training, force_load_plans, num_epochs, prev_trainer_path, call_for_eval)
        loss_base = DC_and_CE_loss({'batch_dice': self.batch_dice, 'smooth': 1e-5, 'do_bg': False}, {})
        self.loss_plop = PODLoss(loss_base, self.ds_loss_weights, self.pod_lambda, self.scales)

    def run_training(self, task, output_folder=None, build_folder=True):
        if not self.was_initialized:
            self.initialize(True, num_epochs=self.max_num_epochs)
        if str(task) not in self.mh_network.heads:
            self._snapshot_old()                        # POD.py:66
        return nnUNetTrainerMultiHead.run_training(self, task, output_folder, build_folder)

    def run_iteration(self, data_generator, do_backprop=True, run_online_evaluation=False, detach=True, no_loss=False):
        return super().run_iteration(data_generator, do_backprop, run_online_evaluation, detach, no_loss, pod=True)
