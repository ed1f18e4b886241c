"""``nnUNetTrainerPLOP`` -- pseudo-labelling of the background with the previous model + local POD distillation.

Mirror of nnunet_ext/training/network_training/plop/nnUNetTrainerPLOP.py: constructor :23-75 (``HYPERPARAMS`` :21),
``initialize`` :77-97 (``loss_orig`` / ``loss_plop``), ``reinitialize`` :99-112, ``extract_max_entropy_and_thresholds``
:114-172, ``run_training`` :174-209 (snapshot of the network as ``network_old`` when a NEW task starts), ``run_iteration``
:211-333 (first task and validation: the original loss; otherwise forward of the current and the old model on the same
batch and the PLOP / POD loss), ``register_forward_hooks`` :335-358.

The reference's forward hooks on every ``conv.Conv*`` module become ``engine.conv_outputs()``: strided views of the
conv / transposed-conv output buffers the HIP engine keeps anyway (no copies, 288 GB of HBM hold both networks' full
activation sets).  The hooks store DETACHED outputs for both models, so POD is a value-only device reduction
(``lnn_local_pod``, one launch per layer); the pseudo-label CE is ``lnn_plop_pseudo_labels`` + two fused CE launches
per weighted level.

Reference behaviour kept in parity mode (every item is pinned by tests/golden/plop_reference.json, which the reference's
own code produced):
  * ``extract_max_entropy_and_thresholds`` compares a Python LIST of label tensors with 0 (PLOP.py:147) -> the background
    mask is ``False``, the histograms stay empty, every threshold is the 0.001 floor (:167-169).  The old model's
    predictions therefore cannot influence the result and its forwards are not run here; the ``num_batches_per_epoch``
    batches are still drawn from the generator of the task trained BEFORE (the new task's loaders are created later, in
    ``super().run_training``), because that shifts the data order;
  * from the THIRD task on the old model is a deepcopy of a network that already carries the current-model hooks
    (registered at the first PLOP iteration of the second task, PLOP.py:227), so its forward overwrites
    ``interm_results`` with its own activations and the POD term is exactly 0.  ``reference_hook_aliasing=False`` is the
    fix behind a flag;
  * thresholds / max_entropy are reset after every ``run_training`` (:205-206).
"""

import torch

from ....losses import MultipleOutputLossPLOP as PLOPLoss
from ..multihead.nnUNetTrainerMultiHead import nnUNetTrainerMultiHead

HYPERPARAMS = {'pod_lambda': float, 'pod_scales': int}


class nnUNetTrainerPLOP(nnUNetTrainerMultiHead):
    def __init__(self, split, task, *args, pod_lambda=1e-2, pod_scales=3, reference_hook_aliasing=True, **kwargs):
        kwargs.setdefault("extension", "plop")
        super().__init__(split, task, *args, **kwargs)
        self.pod_lambda, self.scales = pod_lambda, pod_scales
        self.reference_hook_aliasing = reference_hook_aliasing
        self.network_old = None
        self.old_interm_results, self.interm_results = dict(), dict()
        self.thresholds, self.max_entropy = None, dict()          # PLOP.py:74 (the order of the reference's constructor)
        self.switched = False
        self._taps_on_network = False         # "hooks registered on self.network" (persist: the model object is one)
        self._old_carries_new_taps = False
        self._x_o = None

    def initialize(self, training=True, force_load_plans=False, num_epochs=500, prev_trainer_path=None,
                   call_for_eval=False):
        super().initialize(training, force_load_plans, num_epochs, prev_trainer_path, call_for_eval)
        self.loss_orig = self.loss
        self.loss_plop = PLOPLoss(self.num_classes - 1, self.pod_lambda, self.scales, self.ds_loss_weights)

    def extract_max_entropy_and_thresholds(self):
        """PLOP.py:114-172 as the code runs (see the module docstring): log(K), the 0.001 floor for every level and class,
        ``num_batches_per_epoch`` batches drawn from the current ``tr_gen``."""
        self.max_entropy = float(torch.log(torch.tensor(float(self.num_classes))))
        if self.thresholds is None:
            raise TypeError("'NoneType' object does not support item assignment (thresholds is None until a first "
                            "run_training has finished, PLOP.py:74,156,206)")
        for _ in range(self.num_batches_per_epoch):
            next(self.tr_gen)
        base_threshold = 0.001
        for idx in range(self.plans["num_pool"]):
            self.thresholds[idx] = torch.full((self.num_classes,), base_threshold, dtype=torch.float32, device=self.device)

    def _snapshot_old(self):
        self._old_carries_new_taps = self._taps_on_network          # deepcopy copies the registered hooks too
        self.network_old = self.frozen_copy_of_network()            # PLOP.py:181

    def run_training(self, task, output_folder=None, build_folder=True):
        if not self.was_initialized:
            self.initialize(True, num_epochs=self.max_num_epochs)
        if str(task) not in self.mh_network.heads:
            self._snapshot_old()
            self.extract_max_entropy_and_thresholds()               # PLOP.py:188
        if len(self.mh_network.heads) > 1 and (self.max_entropy is None or not self.thresholds):
            self.extract_max_entropy_and_thresholds()               # restore case, PLOP.py:194-196
        ret = super().run_training(task, output_folder, build_folder)
        self.max_entropy, self.thresholds = None, dict()            # PLOP.py:205-206
        return ret

    def on_forward_done(self, data, output, do_backprop):
        if not self._use_plop:
            return
        outs = output if isinstance(output, (tuple, list)) else (output,)
        interm = self.network.engine_for(data).conv_outputs([o.detach() for o in outs[::-1]])
        with torch.no_grad():
            out_o = self.network_old(data)
            out_o = tuple(out_o) if isinstance(out_o, (tuple, list)) else (out_o,)
            old = self.network_old.engine_for(data).conv_outputs(list(out_o[::-1]))
        if self._old_carries_new_taps and self.reference_hook_aliasing:
            interm = old
        self.old_interm_results, self.interm_results, self._x_o = old, interm, out_o
        if self._pod:
            self.loss_plop.update_plop_params(old, interm)
        else:
            self.loss_plop.update_plop_params(old, interm, self.thresholds, self.max_entropy)

    def run_iteration(self, data_generator, do_backprop=True, run_online_evaluation=False, detach=True, no_loss=False,
                      pod=False):
        first_task = str(self.task) in self.mh_network.heads and len(self.mh_network.heads) == 1
        self._use_plop = not (first_task or run_online_evaluation)
        self._pod = pod
        if self._use_plop:
            assert self.network_old is not None, "no previous model: run_training(task) creates it when a new task starts"
            if pod:
                self.loss = self.loss_plop
            else:
                self.loss = lambda output, target: self.loss_plop(output, self._x_o, target)
            self.switched = True
            self._taps_on_network = True
        else:
            self.loss = self.loss_orig
            self.switched = False
        try:
            return super().run_iteration(data_generator, do_backprop, run_online_evaluation, detach, no_loss)
        finally:
            self._x_o = None
            self.old_interm_results, self.interm_results = dict(), dict()     # PLOP.py:328-329
