"""``nnUNetTrainerEWC`` -- Elastic Weight Consolidation on the flat parameter arena.

Mirror of nnunet_ext/training/network_training/ewc/nnUNetTrainerEWC.py: ``initialize`` :98-140 (EWC loss with
the network's ``named_parameters()`` GENERATOR), ``reinitialize`` :142-177, ``run_training`` :179-230,
``run_iteration`` :232-250 (fresh generator after every iteration), ``after_train`` :252-310; the side data of a finished task
(``ewc_data/fisher_values.pkl`` / ``param_values.pkl``, ``fisher_at`` / ``params_at`` in ``already_trained_on``) is written at
the end of ``run_training`` :205-228 and read back by the constructor :66-78 and by ``initialize(prev_trainer_path=...)``
:104-115, so that a trainer restored from a checkpoint keeps regularising.

Reference behaviours reproduced in parity mode:
  * the penalty covers only the FIRST previous task (generator exhausted by the outer task loop);
  * "Fisher" is the squared gradient of the LAST after_train batch (zero_grad inside the loop), and the
    penalty is absent from that gradient when there are >= 2 batches (generator exhausted on batch 0);
  * a parameter whose ``.grad`` is None (the zero-weight deep-supervision head) gets Fisher ``tensor([1])``.
The gradient is taken UNSCALED by default (the reference's fp32 / CPU path, and what its docstrings describe).  Its
fp16 path squares ``param.grad`` right after ``amp_grad_scaler.scale(loss).backward()`` (EWC.py:287,303) without
``unscale_``: the stored Fisher is (loss_scale * g)^2, i.e. the same ``ewc_lambda`` regularises loss_scale^2
(4.3e9 at the initial 65536) times harder and the value depends on the scaler's history.
``fisher_keeps_loss_scale=True`` reproduces that, so that Fisher pickles of reference fp16 runs are comparable.
``fisher_mode='accumulate'`` is the optional true empirical Fisher (mean of g^2 over the batches, all-reduced
across ranks once per task).
"""
import os
from collections import OrderedDict

import torch

from .... import native as nat
from ....losses import DC_and_CE_loss, MultipleOutputLossEWC as EWCLoss
from ....parallel import all_reduce_stats
from ..multihead.nnUNetTrainerMultiHead import nnUNetTrainerMultiHead

HYPERPARAMS = {'ewc_lambda': float}


class nnUNetTrainerEWC(nnUNetTrainerMultiHead):
    def __init__(self, split, task, *args, ewc_lambda=0.4, fisher_mode="last_batch", fisher_keeps_loss_scale=False,
                 **kwargs):
        kwargs.setdefault("extension", "ewc")
        super().__init__(split, task, *args, **kwargs)
        self.ewc_lambda = ewc_lambda
        self.fisher_mode = fisher_mode
        self.fisher_keeps_loss_scale = fisher_keeps_loss_scale
        # EWC.py:40-58: the method's entries of the fold's ``already_trained_on`` record
        fold = self.already_trained_on.setdefault(str(self.fold), {})
        fold.setdefault('used_ewc_lambda', self.ewc_lambda)
        fold.setdefault('fisher_at', None)
        fold.setdefault('params_at', None)
        # EWC.py:66-78: empty dictionaries, or what an earlier run of this trainer left on disk
        self.fisher = OrderedDict()
        self.params = OrderedDict()
        self._load_fisher_and_params()
        # EWC.py:96
        self.ewc_data_path = None if self.trained_on_path is None else os.path.join(self.trained_on_path, 'ewc_data')

    def _load_fisher_and_params(self):
        fold = self.already_trained_on[str(self.fold)]
        if fold.get('fisher_at') is None or fold.get('params_at') is None:
            return False
        self.fisher = self._load_side_data(fold['fisher_at'])
        self.params = self._load_side_data(fold['params_at'])
        return True

    def initialize(self, training=True, force_load_plans=False, num_epochs=500, prev_trainer_path=None,
                   call_for_eval=False):
        super().initialize(training, force_load_plans, num_epochs, prev_trainer_path, call_for_eval)
        if prev_trainer_path is not None:            # EWC.py:104-115
            self._load_fisher_and_params()
        self.loss = DC_and_CE_loss({'batch_dice': self.batch_dice, 'smooth': 1e-5, 'do_bg': False}, {})
        self.loss = EWCLoss(self.loss, self.ds_loss_weights, self.ewc_lambda, self.fisher, self.params,
                            self.network.named_parameters())

    def reinitialize(self, task, print_loss_info=True):
        super().reinitialize(task, print_loss_info)
        self.loss.update_ewc_params(self.fisher, self.params)

    def run_training(self, task, output_folder=None, build_folder=True):
        if len(self.mh_network.heads) > 0 and str(task) not in self.mh_network.heads:
            assert len(self.fisher) == len(self.mh_network.heads) and len(self.params) == len(self.mh_network.heads), \
                "The number of tasks in the fisher/param values are not as expected --> should be the same as in the Multi Head network."
        if self.task != task or self.tr_gen is None:
            self.reinitialize(task)
        ret = super().run_training(task, output_folder, build_folder)
        self.fisher[task] = OrderedDict()
        self.params[task] = OrderedDict()
        self.after_train()
        self.save_fisher_and_params()
        return ret

    def save_fisher_and_params(self):
        """EWC.py:205-228: dump both dictionaries, record where in ``already_trained_on`` (first time only), rewrite the
        ``<ext>_trained_on.pkl`` file and the ``.pkl`` next to the final checkpoint so that a restore finds them."""
        if self.ewc_data_path is None:
            return
        os.makedirs(self.ewc_data_path, exist_ok=True)
        self._dump_side_data(os.path.join(self.ewc_data_path, 'fisher_values.pkl'), self.fisher)
        self._dump_side_data(os.path.join(self.ewc_data_path, 'param_values.pkl'), self.params)
        fold = self.already_trained_on[str(self.fold)]
        if fold['fisher_at'] is None or fold['params_at'] is None:
            fold['fisher_at'] = os.path.join(self.ewc_data_path, 'fisher_values.pkl')
            fold['params_at'] = os.path.join(self.ewc_data_path, 'param_values.pkl')
            self._write_trained_on_file()
            self.update_init_args()
            if self.output_folder is not None:
                os.makedirs(self.output_folder, exist_ok=True)
                self.save_init_args(os.path.join(self.output_folder, "model_final_checkpoint.model"))

    def run_iteration(self, data_generator, do_backprop=True, run_online_evaluation=False, detach=True, no_loss=False):
        loss = super().run_iteration(data_generator, do_backprop, run_online_evaluation, detach, no_loss)
        self.loss.update_network_params(self.network.named_parameters())
        return loss

    def after_train(self):
        """EWC.py:252-310.  The reference runs ``num_batches_per_epoch`` forward+backward passes with
        ``zero_grad`` before each and no optimiser step, so only the LAST batch's gradient survives; nothing in
        those passes changes state (no step, InstanceNorm keeps no running statistics).  Parity mode therefore
        draws and discards the first n-1 batches and runs ONE forward+backward on the last -- bit-identical
        Fisher for 1/n of the work.  The penalty's presence follows the generator: consumed on the first batch,
        hence absent on the last whenever n >= 2."""
        self.network.train()
        self.optimizer.zero_grad()
        n = self.num_batches_per_epoch
        arena = self.network.arena
        scale = self.amp_grad_scaler.get_scale()
        unscale = 1.0 if self.fisher_keeps_loss_scale else 1.0 / scale
        world_avg = self.dp.averaging_factor if self.dp is not None else 1.0
        if self.fisher_mode == "accumulate":
            facc = torch.zeros_like(arena.grad)
        for b in range(n):
            last = b == n - 1
            if self.fisher_mode != "accumulate" and not last:
                next(self.tr_gen)                           # discarded pass of the reference
                if b == 0 and len(self.loss.tasks) > 0:
                    list(self.loss.network_params)          # ... which would have exhausted the generator (the task
                                                            # loop of DS.py:65-66 only runs when previous tasks exist)
                continue
            self.optimizer.zero_grad()
            data_dict = next(self.tr_gen)
            data = torch.as_tensor(data_dict['data']).to(self.device)
            target = [torch.as_tensor(t).to(self.device) for t in data_dict['target']]
            output = self.network(data)
            loss = self.loss(output, target)
            # parity mode squares the ALL-REDUCED gradient (exchanged during this backward like a training iteration's);
            # accumulate mode squares each rank's OWN gradient: no exchange
            exchange = self.dp is not None and self.fisher_mode != "accumulate"
            if exchange:
                self.dp.begin()
                self.network.on_grad_progress = self.dp.progress
            try:
                self.amp_grad_scaler.scale(loss).backward()
            finally:
                self.network.on_grad_progress = None
            if exchange:
                self.dp.finish()
            if self.fisher_mode == "accumulate":
                nat.call("lnn_fisher_accumulate", arena.grad, facc, arena.size, unscale, 1.0 / n)
        if self.fisher_mode == "accumulate":
            all_reduce_stats(facc, self.process_group)
            fflat = facc * world_avg
        else:
            fflat = torch.empty_like(arena.grad)
            nat.call("lnn_fisher_square", arena.grad, fflat, arena.size, world_avg * unscale)
        no_grad = self.network.params_without_grad
        for name, param in self.network.named_parameters():
            s = param._lnn_slot
            if name in no_grad or not param.requires_grad:
                self.fisher[self.task][name] = torch.tensor([1.0], device=self.device)     # EWC.py:300-301
            else:
                self.fisher[self.task][name] = fflat[s.offset:s.offset + s.numel].view(s.shape).clone()
            self.params[self.task][name] = param.data.clone()
        self.optimizer.zero_grad()
