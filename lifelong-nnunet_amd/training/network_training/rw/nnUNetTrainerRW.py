"""``nnUNetTrainerRW`` -- Riemannian Walk (EWC++ online Fisher + path-integral importance scores) on the flat arenas.

Mirror of nnunet_ext/training/network_training/rw/nnUNetTrainerRW.py: constructor :23-97 (``HYPERPARAMS`` :21),
``initialize`` :99-125, ``reinitialize`` :127-148, ``run_training`` :150-208, ``run_iteration`` :218-229,
``_update_f_s_values`` :231-265 (one fused launch per arena range: ``lnn_rw_update``), ``_extract_params`` :317-322.

Reference behaviours reproduced in parity mode:
  * the statistics are updated after EVERY ``run_iteration`` -- validation ones included, where the gradient is the
    zero left by ``optimizer.zero_grad()``: the Fisher then just decays by (1 - alpha) and ``prev_param`` is re-snapped;
  * the gradient entering the statistics is the unscaled, norm-clipped one (``param.grad`` after MH.py:626-631);
  * after a task: Fisher min-max normalised with the extrema of the per-tensor maxima of the SCORES (:184-188), scores
    scaled to [0, 2] for the first task, untouched for the second, averaged with themselves from the third on (:190-204);
  * the loss receives ``network.named_parameters()`` once (:122-125) and it is never refreshed, so the penalty acts on
    one forward per trainer lifetime (``refresh_network_params=True`` hands over a fresh generator after every iteration,
    as the EWC trainer does).  Value semantics only: the reference additionally iterates the parameters of the network
    object that existed at ``initialize`` time (a deepcopy generation behind the trained one).
"""
import os
from collections import OrderedDict

import torch

from .... import native as nat
from ....losses import DC_and_CE_loss, MultipleOutputLossRW as RWLoss
from ....optim import _Off, _ranges
from ..multihead.nnUNetTrainerMultiHead import nnUNetTrainerMultiHead

HYPERPARAMS = {'rw_alpha': float, 'rw_lambda': float, 'fisher_update_after': int}
EPSILON = 1e-8          # rw/nnUNetTrainerRW.py:17


class nnUNetTrainerRW(nnUNetTrainerMultiHead):
    def __init__(self, split, task, *args, fisher_update_after=10, rw_alpha=0.9, rw_lambda=0.4,
                 refresh_network_params=False, **kwargs):
        kwargs.setdefault("extension", "rw")
        super().__init__(split, task, *args, **kwargs)
        self.alpha, self.rw_lambda, self.fisher_update_after = rw_alpha, rw_lambda, fisher_update_after
        assert self.alpha > 0 and self.alpha <= 1, "rw_alpha should be between 0 and 1: [0, 1]."
        self.refresh_network_params = refresh_network_params
        # RW.py:46-68: the method's entries of the fold's ``already_trained_on`` record
        fold = self.already_trained_on.setdefault(str(self.fold), {})
        fold.setdefault('used_alpha', self.alpha)
        fold.setdefault('used_rw_lambda', self.rw_lambda)
        fold.setdefault('update_fisher_after', self.fisher_update_after)
        for key in ('fisher_at', 'params_at', 'scores_at'):
            fold.setdefault(key, None)
        # RW.py:77-84: empty dictionaries, or what an earlier run of this trainer left on disk
        self.fisher, self.params, self.scores = OrderedDict(), OrderedDict(), OrderedDict()
        self._load_f_p_s_values()
        self.rw_data_path = None if self.trained_on_path is None else os.path.join(self.trained_on_path, 'rw_data')
        self.prev_param, self.count = None, 0
        self._f_flat = self._s_flat = self._prev_flat = None

    def _load_f_p_s_values(self):
        fold = self.already_trained_on[str(self.fold)]
        if any(fold.get(k) is None for k in ('fisher_at', 'params_at', 'scores_at')):
            return False
        self.fisher = self._load_side_data(fold['fisher_at'])
        self.params = self._load_side_data(fold['params_at'])
        self.scores = self._load_side_data(fold['scores_at'])
        return True

    def save_f_p_s_values(self):
        """RW.py:266-300: dump the three dictionaries and record where (first time only)."""
        if self.rw_data_path is None:
            return
        os.makedirs(self.rw_data_path, exist_ok=True)
        names = {'fisher_at': 'fisher_values.pkl', 'params_at': 'param_values.pkl', 'scores_at': 'score_values.pkl'}
        for key, d in (('fisher_at', self.fisher), ('params_at', self.params), ('scores_at', self.scores)):
            self._dump_side_data(os.path.join(self.rw_data_path, names[key]), d)
        fold = self.already_trained_on[str(self.fold)]
        if any(fold[k] is None for k in names):
            for k, n in names.items():
                fold[k] = os.path.join(self.rw_data_path, n)
            self._write_trained_on_file()
            self.update_init_args()
            if self.output_folder is not None:
                os.makedirs(self.output_folder, exist_ok=True)
                self.save_init_args(os.path.join(self.output_folder, "model_final_checkpoint.model"))

    def initialize(self, training=True, force_load_plans=False, num_epochs=500, prev_trainer_path=None,
                   call_for_eval=False):
        super().initialize(training, force_load_plans, num_epochs, prev_trainer_path, call_for_eval)
        if prev_trainer_path is not None:            # RW.py:101-106
            self._load_f_p_s_values()
        assert self.fisher_update_after < self.num_batches_per_epoch, \
            "How should the fisher values and importance scores be calculated if update_after is greater than the number of iterations per epochs.."
        self.loss = DC_and_CE_loss({'batch_dice': self.batch_dice, 'smooth': 1e-5, 'do_bg': False}, {})
        self.loss = RWLoss(self.loss, self.ds_loss_weights, self.rw_lambda, self.fisher, self.params, self.scores,
                           self.network.named_parameters())

    def reinitialize(self, task, print_loss_info=True):
        super().reinitialize(task, print_loss_info)
        self.loss.update_rw_params(self.fisher, self.params, self.scores)

    def run_training(self, task, output_folder=None, build_folder=True):
        if len(self.mh_network.heads) > 0 and str(task) not in self.mh_network.heads:
            assert len(self.fisher) == len(self.mh_network.heads) and len(self.params) == len(self.mh_network.heads), \
                "The number of tasks in the fisher/param values are not as expected --> should be the same as in the Multi Head network."
        if not self.was_initialized:
            self.initialize(True, num_epochs=self.max_num_epochs)
        self.params[task] = OrderedDict()
        # zero Fisher / scores per trainable parameter (:163-169) -- as VIEWS into two flat arenas the fused kernel updates
        arena = self.network.arena
        self._f_flat, self._s_flat = torch.zeros_like(arena.theta), torch.zeros_like(arena.theta)
        self._prev_flat = torch.zeros_like(arena.theta)
        self.fisher[task], self.scores[task] = OrderedDict(), OrderedDict()
        for n, p in self.network.named_parameters():
            if p.requires_grad:
                s = p._lnn_slot
                self.fisher[task][n] = self._f_flat[s.offset:s.offset + s.numel].view(s.shape)
                self.scores[task][n] = self._s_flat[s.offset:s.offset + s.numel].view(s.shape)
        ret = super().run_training(task, output_folder, build_folder)
        self.prev_param, self.count = None, 0
        self._extract_params()
        # -- min-max normalisation (:183-204); extrema of the per-tensor maxima of the SCORES for both -- #
        values = torch.stack([torch.max(v) for v in self.scores[self.task].values()])
        minim, maxim = values.min(), values.max()
        for k, v in list(self.fisher[self.task].items()):
            self.fisher[self.task][k] = (v - minim) / (maxim - minim + EPSILON)
        n_finished = len(self.already_trained_on[str(self.fold)]['finished_training_on'])
        if n_finished == 1:
            for k, v in list(self.scores[self.task].items()):
                self.scores[self.task][k] = 2 * ((v - minim) / (maxim - minim + EPSILON))
        elif n_finished > 2:
            last = self.already_trained_on[str(self.fold)]['finished_training_on'][-1]     # == the task itself (:199)
            prev_scores = {k: v.clone() for k, v in self.scores[last].items()}
            for k, v in list(self.scores[self.task].items()):
                self.scores[self.task][k] = 0.5 * (prev_scores[k] + (v - minim) / (maxim - minim + EPSILON))
        self.save_f_p_s_values()                     # RW.py:206
        return ret

    def run_iteration(self, data_generator, do_backprop=True, run_online_evaluation=False, detach=True, no_loss=False):
        loss = super().run_iteration(data_generator, do_backprop, run_online_evaluation, detach, no_loss)
        self._update_f_s_values()
        if self.refresh_network_params:
            self.loss.update_network_params(self.network.named_parameters())
        return loss

    def _update_f_s_values(self):
        """:231-265 as one launch per arena range of the parameters that have a gradient."""
        if self._f_flat is None:
            return
        if self.count % self.fisher_update_after == 0:
            a = self.network.arena
            inv = getattr(self, "last_inv_scale", 1.0)
            for lo, hi in _ranges(self.network):
                nat.call("lnn_rw_update", _Off(a.theta, lo), _Off(self._prev_flat, lo), _Off(a.grad, lo), _Off(self._f_flat, lo),
                         _Off(self._s_flat, lo), hi - lo, float(inv), 12.0, self.optimizer.ctrl, float(self.alpha), EPSILON,
                         1 if self.prev_param is not None else 0)
            self.prev_param = self._prev_flat
        self.count += 1

    def _extract_params(self):
        for name, param in self.network.named_parameters():
            self.params[self.task][name] = param.data.clone()
