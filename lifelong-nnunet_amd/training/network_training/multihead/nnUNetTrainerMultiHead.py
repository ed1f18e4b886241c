"""``nnUNetTrainerMultiHead`` -- base trainer of every continual-learning method, hot-path part only.

Host-side mirror of nnunet_ext/training/network_training/multihead/nnUNetTrainerMultiHead.py:
  ctor kwargs :40-44, ``initialize`` :303, ``initialize_optimizer_and_scheduler`` :294-301,
  ``reinitialize`` :458, ``run_training`` :520-596, ``run_iteration`` :598-656, ``on_epoch_end`` :658,
  ``run_online_evaluation`` :924-961, ``finish_online_evaluation_extended`` :963-1049,
  ``save_checkpoint`` / ``load_checkpoint_ram`` :1164-1313, ``_update_loss_after_plans_change`` :1363-1387,
  ``reorder_UNet_components`` :1391-1408.
Out of scope here (CLI, plans files, NIfTI I/O, progress plots): network topology comes
from a small ``plans`` dict instead of a pickled plans file, data from a ``data_provider`` callable yielding
the reference's data dicts ``{'data','target','keys'}`` (MH.py:606-608).
"""
from __future__ import annotations

import warnings
import os
from collections import OrderedDict
from typing import Callable, Optional

import numpy as np
import torch

from ....losses import DC_and_CE_loss, MultipleOutputLoss2, ds_loss_weights
from ....multihead import MultiHead_Module
from ....network import Generic_UNet
from ....optim import DeferredLoss, FusedSGD, GradScaler
from ....parallel import GradAllReducer
from ....synthetic import SyntheticPatchGenerator

HYPERPARAMS = {}

DEFAULT_PLANS = {  # Task004_Hippocampus 3d_fullres shaped (BASELINE.json configs[0]); B=2 for plumbing
    "patch_size": (40, 56, 40), "batch_size": 2, "num_pool": 3, "base_num_features": 32,
    "num_classes": 3, "num_input_channels": 1,
}


def default_data_provider(task, split, plans, seed=12345):
    """Deterministic synthetic patches per (task, split); tasks differ by seed and blob scale."""
    h = sum(ord(c) * (i + 1) for i, c in enumerate(str(task))) % 9973
    return SyntheticPatchGenerator(plans["batch_size"], plans["patch_size"], plans["num_pool"],
                                   plans["num_input_channels"], plans["num_classes"],
                                   seed=seed + 17 * h + (0 if split == "train" else 500000),
                                   period=plans.get("synthetic_period", 4),
                                   blob_scale=1.0 + 0.5 * (h % 3), key_prefix=f"{task}_{split}",
                                   pool_op_kernel_sizes=plans.get("pool_op_kernel_sizes"))


class nnUNetTrainerMultiHead:
    def __init__(self, split, task, plans_file=None, fold=0, output_folder=None, dataset_directory=None,
                 batch_dice=False, stage=None, unpack_data=True, deterministic=True, fp16=True, save_interval=5,
                 already_trained_on=None, use_progress=True, identifier="lnn_amd", extension='multihead',
                 tasks_list_with_char=None, mixed_precision=True, save_csv=True, del_log=False, use_vit=False,
                 vit_type='base', version=1, split_gpu=False, transfer_heads=False, ViT_task_specific_ln=False,
                 do_LSA=False, do_SPT=False, network=None, use_param_split=False,
                 plans: Optional[dict] = None, data_provider: Optional[Callable] = None, device="cuda",
                 process_group=None, deterministic_wgrad=False):
        assert not use_vit, "Generic_ViT_UNet variants are out of scope (SURVEY.md section 2 row 8)"
        # the positional constructor arguments the reference stores next to every checkpoint (MH.py:181-185)
        self.init_args = (split, task, plans_file, fold, output_folder, dataset_directory, batch_dice, stage, unpack_data,
                          deterministic, fp16, save_interval, already_trained_on, use_progress, identifier, extension,
                          tasks_list_with_char, mixed_precision, save_csv, del_log, use_vit, vit_type, version, split_gpu,
                          transfer_heads, ViT_task_specific_ln, do_LSA, do_SPT)
        self.split, self.task, self.fold = split, task, fold
        self.plans = dict(DEFAULT_PLANS if plans is None else plans)
        self.num_classes = self.plans["num_classes"]      # logit channels, background included (upstream process_plans)
        self.batch_dice = batch_dice          # False for single-stage 3d_fullres (run/default_configuration.py:93-100)
        self.deterministic, self.fp16 = deterministic, fp16
        self.save_interval, self.extension = save_interval, extension
        self.transfer_heads = transfer_heads
        self.device = torch.device(device)
        self.data_provider = data_provider or default_data_provider
        self.process_group = process_group
        # bit-reproducible fp16 steps: ordered reductions instead of fp32 atomics in the weight-gradient kernels (what
        # torch.backends.cudnn.deterministic, which upstream's ``deterministic`` flag sets, stands for); off by default
        self.deterministic_wgrad = deterministic_wgrad
        self.already_trained_on = already_trained_on or OrderedDict()
        self.tasks_list_with_char = tasks_list_with_char
        # MH.py:125-129: ``<extension>_trained_on.pkl`` and the side data of the regularising trainers (``ewc_data/``, ``rw_data/``)
        # live two levels above the fold's output folder.  (The reference inserts its task-sequence folders there through
        # ``_build_output_path``; that directory layout belongs to its run scripts, SURVEY.md section 2 "side", and a caller who wants
        # it sets ``trained_on_path`` after construction.)  Without an output folder nothing is written.
        self.output_folder = output_folder
        self.trained_on_path = None if output_folder is None else \
            os.path.dirname(os.path.dirname(os.path.realpath(output_folder)))
        # upstream nnUNetTrainerV2 constants
        self.initial_lr, self.weight_decay = 1e-2, 3e-5
        self.max_grad_norm = 12.0       # clip_grad_norm_(parameters, 12) of the iteration (MH.py:629,640); None: no clipping
        self.max_num_epochs = 500
        self.num_batches_per_epoch, self.num_val_batches_per_epoch = 250, 50
        self.epoch = 0
        self.was_initialized = False
        self.network = self.mh_network = self.optimizer = self.loss = None
        self.tr_gen = self.val_gen = None
        self.trainer_model = network
        self.online_eval_foreground_dc, self.online_eval_tp, self.online_eval_fp, self.online_eval_fn = [], [], [], []
        self.subject_names_raw = []
        self._last_eval_keys = None
        self.eval_names_from_same_batch = False       # False = the reference's tee() pairing in _perform_validation
        self.all_tr_losses, self.all_val_losses = [], []
        self.validation_results = dict()
        self.amp_grad_scaler = None
        self.dp: Optional[GradAllReducer] = None
        self._last_grad_norm, self._last_found_inf = None, False
        # run_iteration(detach=True) returns the loss as a number on the host (MH.py:655).  With ``defer_loss_fetch`` it returns a
        # DeferredLoss instead: the device-to-host copy of {loss, gradient norm, found-inf} is in flight and the GradScaler update of
        # THIS iteration is applied when the next one asks for the scale -- the host never waits for the device between two
        # iterations.  The epoch loop of run_training switches it on (it only needs the numbers at the end of the epoch).
        self.defer_loss_fetch = False
        self._pending_step = None
        if deterministic:
            torch.manual_seed(12345 + fold)      # the only seed constants the reference uses (MH.py:214,259)
            np.random.seed(12345 + fold)

    # ------------------------------------------------------------------------------------------ init
    def initialize(self, training=True, force_load_plans=False, num_epochs=500, prev_trainer_path=None,
                   call_for_eval=False):
        self.max_num_epochs = num_epochs
        self.initialize_network()
        self._update_loss_after_plans_change(self.plans.get("pool_op_kernel_sizes") or self.plans["num_pool"], self.plans["patch_size"])
        if training:
            self.tr_gen = self.data_provider(self.task, "train", self.plans)
            self.val_gen = self.data_provider(self.task, "val", self.plans)
        self.initialize_optimizer_and_scheduler()
        # fp16=False (MH.py:632-641, --fp32): fp32 activation storage on the direct fp32 kernels, no loss scaling
        self.network.storage = "fp16" if self.fp16 else "fp32"
        self.amp_grad_scaler = GradScaler(enabled=bool(self.fp16))
        import os
        force_dp = os.environ.get("LNN_FORCE_DP", "0") == "1"
        if torch.distributed.is_available() and torch.distributed.is_initialized() and \
                (torch.distributed.get_world_size(self.process_group) > 1 or force_dp):
            self.dp = GradAllReducer(self.network.arena.grad, self.process_group, force=force_dp)
        self.was_initialized = True

    def initialize_network(self):
        p = self.plans
        prev = self.trainer_model
        # the network is built from the plans (MH.py:348-369 -> upstream nnUNetTrainerV2.initialize_network): input channels,
        # one pooling per level and one conv kernel per stage; plans without the two lists are isotropic (2x2x2 / 3x3x3)
        if p.get("pool_op_kernel_sizes") is not None:
            assert len(p["pool_op_kernel_sizes"]) == p["num_pool"], "num_pool = len(net_num_pool_op_kernel_sizes) (MH.py:360)"
        self.mh_network = MultiHead_Module(Generic_UNet, self.split, self.task, prev, p["num_input_channels"],
                                           p["base_num_features"], p["num_classes"], p["num_pool"],
                                           device=self.device, pool_op_kernel_sizes=p.get("pool_op_kernel_sizes"),
                                           conv_kernel_sizes=p.get("conv_kernel_sizes"))
        self.network = self.mh_network.model
        self.network.deterministic_wgrad = bool(self.deterministic_wgrad)
        self.network.inference_apply_nonlin = lambda x: torch.softmax(x, 1)

    def initialize_optimizer_and_scheduler(self):
        assert self.network is not None, "self.initialize_network must be called first"
        self.optimizer = FusedSGD(self.network, self.initial_lr, weight_decay=self.weight_decay, momentum=0.99,
                                  nesterov=True)
        self.lr_scheduler = None

    def _update_loss_after_plans_change(self, net_num_pool_op_kernel_sizes, patch_size):
        net_numpool = net_num_pool_op_kernel_sizes if isinstance(net_num_pool_op_kernel_sizes, int) \
            else len(net_num_pool_op_kernel_sizes)
        self.ds_loss_weights = ds_loss_weights(net_numpool)
        self.loss = DC_and_CE_loss({'batch_dice': self.batch_dice, 'smooth': 1e-5, 'do_bg': False}, {})
        self.loss = MultipleOutputLoss2(self.loss, self.ds_loss_weights)

    def reinitialize(self, task, print_loss_info=True):
        """MH.py:458-518: new task -> new generators (the loss object is kept unless a subclass swaps it)."""
        self.task = task
        self.tr_gen = self.data_provider(task, "train", self.plans)
        self.val_gen = self.data_provider(task, "val", self.plans)

    def frozen_copy_of_network(self):
        """``copy.deepcopy(self.network)`` of the distillation trainers (MiB.py:96, PLOP.py:181, POD.py:66): a second network
        object with the current weights (own parameter arena, weight panels and activation buffers), same storage mode,
        evaluated without autograd."""
        from ....network import Generic_UNet
        p = self.plans
        old = Generic_UNet(p["num_input_channels"], p["base_num_features"], p["num_classes"], p["num_pool"], device=self.device,
                           pool_op_kernel_sizes=p.get("pool_op_kernel_sizes"), conv_kernel_sizes=p.get("conv_kernel_sizes"))
        old.storage = self.network.storage
        old.load_state_dict(self.network.state_dict())
        for prm in old.parameters():
            prm.requires_grad = False
        old.eval()
        return old

    def maybe_update_lr(self, epoch=None):
        ep = self.epoch + 1 if epoch is None else epoch
        self.optimizer.param_groups[0]['lr'] = self.initial_lr * (1 - ep / self.max_num_epochs) ** 0.9   # upstream poly_lr

    # ------------------------------------------------------------------------------------------ training
    def run_training(self, task, output_folder=None, build_folder=True):
        """MH.py:520-596: (re)point the generators at ``task``, add / activate its head, run the epoch loop."""
        if not self.was_initialized:
            self.initialize(True, num_epochs=self.max_num_epochs)
        if self.task != task or self.tr_gen is None:
            self.reinitialize(task)
        if str(task) not in self.mh_network.heads:
            self.mh_network.add_new_task(task, use_init=not self.transfer_heads)
        self.network = self.mh_network.assemble_model(task)
        self.already_trained_on.setdefault(str(self.fold), {}).setdefault('finished_training_on', [])
        ret = self._run_epoch_loop()
        self.already_trained_on[str(self.fold)]['finished_training_on'].append(task)
        self.epoch = 0
        return ret

    def _run_epoch_loop(self):
        """upstream NetworkTrainer.run_training: per epoch 250 train iterations, 50 no-grad validation iterations."""
        self.maybe_update_lr(self.epoch)
        while self.epoch < self.max_num_epochs:
            self.network.train()
            # the epoch loop only needs the losses at the end of the epoch (upstream: train_losses_epoch.append(l); np.mean): the
            # iterations run without a host synchronisation between them.  Subclasses whose run_iteration reads host-side state of
            # the step it just enqueued are unaffected: last_grad_norm / last_found_inf resolve on access.
            keep, self.defer_loss_fetch = self.defer_loss_fetch, True
            try:
                tr = [self.run_iteration(self.tr_gen, True) for _ in range(self.num_batches_per_epoch)]
            finally:
                self.defer_loss_fetch = keep
            self._finish_pending_step()
            self.all_tr_losses.append(float(np.mean([float(v) for v in tr])))
            with torch.no_grad():
                self.network.eval()
                va = []
                for _ in range(self.num_val_batches_per_epoch):
                    va.append(self.run_iteration(self.val_gen, False, True))
                    if self._last_eval_keys is not None:       # epoch-end evaluation pairs every batch with its own names
                        self.subject_names_raw.append(np.asarray(self._last_eval_keys))
                if va:
                    self.all_val_losses.append(float(np.mean(va)))
            self.on_epoch_end()
            self.epoch += 1
        return self.all_tr_losses

    def on_epoch_end(self):
        if self.online_eval_tp:
            self.last_online_eval = self.finish_online_evaluation_extended(self.task)
        self.maybe_update_lr()

    def run_iteration(self, data_generator, do_backprop=True, run_online_evaluation=False, detach=True, no_loss=False):
        """MH.py:598-656.  Device work is enqueued without host synchronisation; the single sync is the loss
        fetch at the end (the reference's ``.cpu().numpy()``, MH.py:655), which also carries the gradient-norm
        / found-inf pair the GradScaler update needs."""
        data_dict = next(data_generator)
        data = torch.as_tensor(data_dict['data']).to(self.device, non_blocking=True)
        target = [torch.as_tensor(t).to(self.device, non_blocking=True) for t in data_dict['target']]
        self.optimizer.zero_grad()
        output = self.network(data)
        self.on_forward_done(data, output, do_backprop)
        l = None
        if not no_loss:
            self._bind_process_group(self.loss)
            l = self.loss(output, target)
        if do_backprop:
            self._finish_pending_step()            # the previous iteration's found-inf flag decides this iteration's scale
            scale = self.amp_grad_scaler.get_scale()
            if self.dp is not None:
                self.dp.begin()
                self.network.on_grad_progress = self.dp.progress
            try:
                self.amp_grad_scaler.backward(l)       # = scale(l).backward(), the scale as the seed gradient
            finally:
                # also when backward raises and the caller carries on: a backward outside run_iteration exchanges nothing unless it
                # asks to (a stale callback would all-reduce against the bucket state of THIS iteration)
                self.network.on_grad_progress = None
            world_avg = 1.0
            if self.dp is not None:
                self.dp.finish()
                world_avg = self.dp.averaging_factor
            inv = world_avg / scale
            self.last_inv_scale = inv
            self.optimizer.grad_norm_pass(inv)                    # unscale_ + the norm of clip_grad_norm_(…, 12)
            self.optimizer.step(inv_scale=inv, max_norm=float(self.max_grad_norm or 0.0))   # clip coefficient + inf-skip on device
        if run_online_evaluation:
            self.run_online_evaluation(output, target)
            self._last_eval_keys = data_dict.get('keys')
        if do_backprop:
            self.mh_network.update_after_iteration()
        if no_loss:
            return None
        if detach:
            if do_backprop:
                if self.defer_loss_fetch:
                    self._pending_step = self.optimizer.fetch_with_loss_async(l)
                    return DeferredLoss(self._pending_step)
                vals = self.optimizer.fetch_with_loss(l)          # {sum g^2, #non-finite, loss}: one copy launch + one D2H
                self._last_grad_norm, self._last_found_inf = float(vals[0]) ** 0.5, bool(vals[1] > 0)
                self.amp_grad_scaler.update(self._last_found_inf)
                return np.float32(vals[2])
            return l.detach().cpu().numpy()
        return l

    def _finish_pending_step(self):
        """Bookkeeping of an iteration whose numbers were fetched without a synchronisation (``defer_loss_fetch``): gradient norm,
        found-inf flag, the optimiser's momentum-state record and the GradScaler update (MH.py:631 ``amp_grad_scaler.update()``)."""
        h = self._pending_step
        if h is None:
            return
        self._pending_step = None
        vals = h.get()
        self._last_grad_norm, self._last_found_inf = float(vals[0]) ** 0.5, bool(vals[1] > 0)
        self.optimizer._resolve_pending(self._last_found_inf)
        self.amp_grad_scaler.update(self._last_found_inf)

    @property
    def last_grad_norm(self):
        self._finish_pending_step()
        return self._last_grad_norm

    @last_grad_norm.setter
    def last_grad_norm(self, v):
        self._last_grad_norm = v

    @property
    def last_found_inf(self):
        self._finish_pending_step()
        return self._last_found_inf

    @last_found_inf.setter
    def last_found_inf(self, v):
        self._last_found_inf = v

    def _bind_process_group(self, loss):
        """The batch-Dice exchange inside DC_and_CE_loss must run over THIS trainer's ranks (``process_group``), not the
        default group: hand it to whatever Dice+CE object the (possibly wrapped) loss holds.  Done once per loss object."""
        if getattr(loss, "_lnn_group_bound", None) is self.process_group and hasattr(loss, "_lnn_group_bound"):
            return
        seen, todo = set(), [loss]
        while todo:
            m = todo.pop()
            if id(m) in seen or m is None:
                continue
            seen.add(id(m))
            if isinstance(m, DC_and_CE_loss):
                m.process_group = self.process_group
            for attr in ("loss", "base_loss"):
                todo.append(getattr(m, attr, None))
        try:
            loss._lnn_group_bound = self.process_group
        except AttributeError:
            pass

    def on_forward_done(self, data, output, do_backprop):
        """Hook between forward and loss (used by LwF to read the old heads on the fresh body activations)."""

    # ------------------------------------------------------------------------------------------ evaluation
    def run_online_evaluation(self, output, target):
        """MH.py:924-961: argmax of the full-resolution output, per-sample hard TP/FP/FN per foreground class."""
        from .... import native as nat
        out = output[0] if isinstance(output, (tuple, list)) else output
        tgt = target[0] if isinstance(target, (tuple, list)) else target
        N, K = out.shape[:2]
        V = out[0, 0].numel()
        counts = torch.empty((N, K - 1, 3), device=out.device)
        nat.call("lnn_online_dice_counts", out.contiguous(), tgt.reshape(N, V).float().contiguous(), N, K, V, counts)
        c = counts.cpu().numpy()
        self.online_eval_tp.append(c[:, :, 0]); self.online_eval_fp.append(c[:, :, 1]); self.online_eval_fn.append(c[:, :, 2])

    def finish_online_evaluation_extended(self, task, unique_subject_names=None):
        """MH.py:963-1049.  TP / FP / FN of all samples that carry the same subject name (``self.subject_names_raw``, one
        array of names per evaluated batch, MH.py:802,819) are SUMMED first; Dice = 2TP/(2TP+FP+FN) and
        IoU = TP/(TP+FP+FN) are then taken per subject and foreground class and stored under
        ``self.validation_results['epoch_<e>'][task][subject]['mask_<c>']`` exactly as the reference does (0/0 stays NaN
        in the stored value).  Without recorded names every sample is its own subject.  Returns a summary (means over
        subjects, NaN entries ignored) for callers that want one number; the reference returns nothing."""
        tp = np.array(self.online_eval_tp); fp = np.array(self.online_eval_fp); fn = np.array(self.online_eval_fn)
        tp = tp.reshape(-1, tp.shape[-1]); fp = fp.reshape(-1, fp.shape[-1]); fn = fn.reshape(-1, fn.shape[-1])
        raw = np.array(self.subject_names_raw).flatten()
        if raw.size == 0:
            raw = np.array([f"sample_{i:06d}" for i in range(tp.shape[0])])
        assert raw.shape[0] == tp.shape[0], f"{raw.shape[0]} subject names for {tp.shape[0]} evaluated samples"
        subject_names = list(np.unique(raw)) if unique_subject_names is None else list(unique_subject_names)
        store, dice_rows, iou_rows = dict(), [], []
        for subject in subject_names:
            idx = np.where(raw == subject)
            i, j, k = tp[idx].sum(axis=0), fp[idx].sum(axis=0), fn[idx].sum(axis=0)
            if np.isnan(i).any():                      # MH.py:1015-1022: subjects with NaN counts are dropped
                continue
            with np.errstate(invalid='ignore', divide='ignore'):
                iou, dc = i / (i + j + k), 2 * i / (2 * i + j + k)
            store[str(subject)] = {'mask_' + str(c + 1): {'IoU': np.float64(iou[c]), 'Dice': np.float64(dc[c])}
                                   for c in range(len(iou))}
            dice_rows.append(dc); iou_rows.append(iou)
        self.validation_results.setdefault('epoch_' + str(self.epoch), {})[task] = store
        self.online_eval_foreground_dc, self.online_eval_tp, self.online_eval_fp, self.online_eval_fn = [], [], [], []
        self.subject_names_raw = []
        dice, iou = np.array(dice_rows), np.array(iou_rows)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            return {"mean_dice_per_class": np.nanmean(dice, 0).tolist(), "mean_iou_per_class": np.nanmean(iou, 0).tolist(),
                    "mean_dice": float(np.nanmean(dice)), "mean_iou": float(np.nanmean(iou)), "per_subject": store}

    def _perform_validation(self, use_tasks=None, num_batches=None, call_for_eval=False):
        """MH.py:678-901 hot part: for every head, assemble it, run no-grad iterations on that task's validation
        generator with online evaluation, collect per-subject Dice/IoU.  Subject names: the reference draws them from a
        ``tee(self.val_gen, 1)[0]`` copy (MH.py:812-819), which for a generator ADVANCES the generator itself -- the
        names of batch 2k are paired with the predictions of batch 2k+1.  Parity mode reproduces that;
        ``self.eval_names_from_same_batch = True`` pairs every prediction with its own batch's names."""
        use_tasks = use_tasks or list(self.mh_network.heads.keys())
        num_batches = num_batches or self.num_val_batches_per_epoch
        active = self.mh_network.active_task
        results = {}
        with torch.no_grad():
            for t in use_tasks:
                self.network = self.mh_network.assemble_model(t)
                self.network.eval()
                gen = self.data_provider(t, "val", self.plans)
                for _ in range(num_batches):
                    if getattr(self, "eval_names_from_same_batch", False):
                        batch = next(gen)
                        self.subject_names_raw.append(np.asarray(batch['keys']))
                        self.run_iteration(iter([batch]), False, True, no_loss=call_for_eval)
                    else:
                        self.subject_names_raw.append(np.asarray(next(gen)['keys']))
                        self.run_iteration(gen, False, True, no_loss=call_for_eval)
                results[str(t)] = self.finish_online_evaluation_extended(t)
        self.network = self.mh_network.assemble_model(active)
        self.network.train()
        return results

    # ------------------------------------------------------------------------------------------ persistence
    def predict_preprocessed_data_return_seg_and_softmax(self, data, do_mirroring=True, mirror_axes=None,
                                                         use_sliding_window=True, step_size=0.5, use_gaussian=True,
                                                         pad_border_mode='constant', pad_kwargs=None, all_in_gpu=False,
                                                         verbose=True, mixed_precision=True):
        """Upstream nnUNetTrainer method the reference calls at predict.py:208-219 and through ``validate``
        (MH.py:1115): tiled prediction of one preprocessed case with the CURRENT head -> (segmentation, softmax)."""
        from ....inference import predict_3D
        assert pad_border_mode == 'constant', "the reference only uses constant (zero) padding here"
        if mirror_axes is None:
            mirror_axes = (0, 1, 2)
        return predict_3D(self.network, data, do_mirroring=do_mirroring, mirror_axes=tuple(mirror_axes),
                          use_sliding_window=use_sliding_window, step_size=step_size, patch_size=tuple(self.plans["patch_size"]),
                          use_gaussian=use_gaussian, verbose=verbose)

    def validate(self, do_mirroring=True, use_sliding_window=True, step_size=0.5, save_softmax=True, use_gaussian=True,
                 overwrite=True, validation_folder_name='validation_raw', debug=False, all_in_gpu=False,
                 segmentation_export_kwargs=None, run_postprocessing_on_folds=True, output_folder=None):
        """MH.py:1052-1135 (-> upstream ``nnUNetTrainer.validate`` per task): for EVERY head the model holds, in head order,
        that task's preprocessed dataset is loaded and split, the head is assembled, and every case of the VALIDATION split
        is predicted as a whole volume by tiled inference (``predict_preprocessed_data_return_seg_and_softmax``: sliding
        window, Gaussian importance, test-time mirroring); the per-subject summary is the reference evaluator's dictionary
        (``evaluation.compute_scores_and_build_dict``, evaluator2.py:60-109: IoU / Dice per mask, ``None`` for a class absent
        from both volumes).  The reference writes NIfTI predictions into ``<output_folder>/<validation_folder_name><task>``
        and returns a list of ``None``; NIfTI export is out of scope, so with ``output_folder`` the
        segmentation (and the softmax with ``save_softmax``) of every case goes there as ``<case>.npz`` next to a
        ``summary.json``, and the returned list holds one ``{'task', 'cases', 'summary'}`` entry per head.  Needs a data
        provider over preprocessed folders (``dataset_for`` / ``splits_file_for``, dataloading.PreprocessedDataProvider)."""
        import json
        import os
        from ....dataloading import do_split
        from ....evaluation import compute_scores_and_build_dict, summarize
        trained_on = list(self.mh_network.heads.keys())
        assert len(trained_on) != 0, "Before performing any validation, the model needs to be trained on at least one task."
        prov = self.data_provider
        assert hasattr(prov, "dataset_for") and hasattr(prov, "splits_file_for"), \
            "validate() predicts whole preprocessed cases: the data provider must expose dataset_for(task) / splits_file_for(task)"
        active = self.mh_network.active_task
        ret_joined = list()
        self.network.eval()
        num_fg = self.plans["num_classes"] - 1            # evaluator2.py:61,92: plan['num_classes'] counts foreground classes
        for task in trained_on:
            dataset = prov.dataset_for(task)
            _, dataset_val = do_split(dataset, self.fold, prov.splits_file_for(task))
            self.network = self.mh_network.assemble_model(task)
            self.network.eval()
            folder = None
            if output_folder is not None:
                folder = os.path.join(output_folder, validation_folder_name + str(task))
                os.makedirs(folder, exist_ok=True)
            cases = OrderedDict()
            for k, entry in dataset_val.items():
                fname = None if folder is None else os.path.join(folder, k + ".npz")
                if fname is not None and not overwrite and os.path.isfile(fname):
                    seg = np.load(fname)["seg"]
                else:
                    npy = entry['data_file'][:-4] + ".npy"
                    data = np.load(npy, 'r') if os.path.isfile(npy) else np.load(entry['data_file'])['data']
                    seg, softmax = self.predict_preprocessed_data_return_seg_and_softmax(
                        data[:-1], do_mirroring=do_mirroring, use_sliding_window=use_sliding_window, step_size=step_size,
                        use_gaussian=use_gaussian, all_in_gpu=all_in_gpu, verbose=False, mixed_precision=self.fp16)
                    if fname is not None:
                        if save_softmax:
                            np.savez_compressed(fname, seg=seg.astype(np.uint8), softmax=softmax.astype(np.float16))
                        else:
                            np.savez_compressed(fname, seg=seg.astype(np.uint8))
                gt = np.load(entry['data_file'][:-4] + ".npy", 'r')[-1] if os.path.isfile(entry['data_file'][:-4] + ".npy") \
                    else np.load(entry['data_file'])['data'][-1]
                cases[k] = (seg, np.maximum(np.asarray(gt), 0))          # -1 marks voxels outside the non-zero mask: background
            cases_dict = compute_scores_and_build_dict(cases, num_fg)
            entry = {"task": task, "cases": cases_dict, "summary": summarize(cases_dict)}
            if folder is not None:
                with open(os.path.join(folder, "summary.json"), "w") as f:
                    json.dump(entry, f, indent=1)
            ret_joined.append(entry)
        self.already_trained_on.setdefault(str(self.fold), {}).setdefault('finished_validation_on', []).append(trained_on[-1])
        self.network = self.mh_network.assemble_model(active)
        self.network.train()
        return ret_joined

    # ------------------------------------------------------------------------------------------ side data of a finished task
    def update_init_args(self):
        """MH.py:1199-1208: position 12 of the stored constructor arguments is ``already_trained_on``."""
        init = list(self.init_args)
        init[12] = self.already_trained_on
        self.init_args = tuple(init)

    def save_init_args(self, fname):
        """MH.py:1210-1222: (re)write ``fname + '.pkl'`` -- constructor arguments, class, plans -- next to a checkpoint."""
        import pickle
        info = OrderedDict(init=self.init_args, name=self.__class__.__name__, plans=self.plans)
        info["class"] = str(self.__class__)
        with open(fname + ".pkl", "wb") as f:
            pickle.dump(info, f)

    def _write_trained_on_file(self):
        """``<extension>_trained_on.pkl`` (MH.py:380,872,1125: ``write_pickle``; the EWC / RW / LwF trainers write the same name
        with ``save_json``, EWC.py:224 -- SURVEY.md Appendix C -- and the base class overwrites it with a pickle at the next
        checkpoint: the pickle is what is written here)."""
        if self.trained_on_path is None:
            return
        import pickle
        os.makedirs(self.trained_on_path, exist_ok=True)
        with open(os.path.join(self.trained_on_path, self.extension + '_trained_on.pkl'), "wb") as f:
            pickle.dump(self.already_trained_on, f)

    @staticmethod
    def read_trained_on_file(path):
        """``<extension>_trained_on.pkl`` as a dictionary, whichever of its two writers produced it: ``write_pickle`` (MH.py:380) or
        ``save_json`` under the same name (EWC.py:224, RW.py:296, LWF.py:233)."""
        import json
        import pickle
        with open(path, "rb") as f:
            raw = f.read()
        try:
            return pickle.loads(raw)
        except Exception:
            return json.loads(raw.decode())

    @staticmethod
    def _dump_side_data(path, per_task):
        """``write_pickle`` of a dict task -> parameter name -> tensor (EWC.py:215-216, RW.py:283-285), as CPU tensors: the reference
        calls ``.cpu()`` and drops the result (EWC.py:206-211), so ITS pickles hold device tensors and only load on a machine with
        the same device; both kinds load here."""
        import pickle
        out = OrderedDict((task, OrderedDict((k, v.detach().cpu().clone()) for k, v in d.items())) for task, d in per_task.items())
        with open(path, "wb") as f:
            pickle.dump(out, f)

    def _load_side_data(self, path):
        """``load_pickle`` + ``to_cuda`` (EWC.py:70-78)."""
        import pickle
        with open(path, "rb") as f:
            d = pickle.load(f)
        return OrderedDict((task, OrderedDict((k, torch.as_tensor(v).to(self.device)) for k, v in per.items()))
                           for task, per in d.items())

    def save_checkpoint(self, fname=None, save_optimizer=True):
        """MH.py:1164-1197 -> upstream ``NetworkTrainer.save_checkpoint`` / ``nnUNetTrainer.save_checkpoint``: the WHOLE
        MultiHead_Module state (``model.* / body.* / heads.<task>.*``, CPU tensors), ``epoch + 1``, the optimiser and
        GradScaler state in torch's own layouts, the loss curves, and next to ``fname`` a ``.pkl`` with the constructor
        arguments and plans -- the dictionary a reference trainer's ``load_checkpoint_ram`` consumes.  The head list and
        the active task (kept by the reference in ``<ext>_trained_on.pkl``, MH.py:1174-1180) are stored in the
        checkpoint as well so that it is self-contained."""
        self.already_trained_on.setdefault(str(self.fold), {})
        self.already_trained_on[str(self.fold)]['checkpoint_should_exist'] = True
        self.already_trained_on[str(self.fold)]['tasks_at_time_of_checkpoint'] = list(self.mh_network.heads.keys())
        self.already_trained_on[str(self.fold)]['active_task_at_time_of_checkpoint'] = self.mh_network.active_task
        ckpt = {"epoch": self.epoch + 1,
                "state_dict": OrderedDict((k, v.detach().cpu().clone()) for k, v in self.mh_network.state_dict().items()),
                "optimizer_state_dict": self.optimizer.state_dict() if (save_optimizer and self.optimizer) else None,
                "lr_scheduler_state_dict": None,
                "plot_stuff": (self.all_tr_losses, self.all_val_losses, [], []),
                "best_stuff": (None, None, None),
                "heads": list(self.mh_network.heads.keys()), "active_task": self.mh_network.active_task}
        if ckpt["optimizer_state_dict"] is not None:
            for st in ckpt["optimizer_state_dict"]["state"].values():
                st["momentum_buffer"] = st["momentum_buffer"].cpu()
        if self.amp_grad_scaler is not None:
            ckpt["amp_grad_scaler"] = self.amp_grad_scaler.state_dict()
        if fname is not None:
            import pickle
            torch.save(ckpt, fname)
            info = OrderedDict(init=self.init_args, name=self.__class__.__name__, plans=self.plans)
            info["class"] = str(self.__class__)
            with open(fname + ".pkl", "wb") as f:
                pickle.dump(info, f)
        return ckpt

    def load_checkpoint_ram(self, checkpoint, train=True):
        """MH.py:1278-1313: the heads must exist before the state dict is loaded; ``module.`` prefixes are stripped as
        upstream does; optimiser / GradScaler states are read in torch's layouts (or this package's round-1 one)."""
        heads = checkpoint.get("heads") or self.already_trained_on[str(self.fold)]['tasks_at_time_of_checkpoint']
        active = checkpoint.get("active_task") or self.already_trained_on[str(self.fold)]['active_task_at_time_of_checkpoint']
        self.mh_network.add_n_tasks_and_activate(heads, active)
        curr = set(self.mh_network.state_dict().keys())
        sd = OrderedDict((k[7:] if (k not in curr and k.startswith("module.")) else k, v) for k, v in checkpoint["state_dict"].items())
        self.mh_network.load_state_dict(sd)
        self.network = self.mh_network.model
        self.network.mark_params_changed()
        self.epoch = checkpoint.get("epoch", 0)
        if train and checkpoint.get("optimizer_state_dict") is not None:
            self.optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
        if "plot_stuff" in checkpoint:
            self.all_tr_losses, self.all_val_losses = list(checkpoint["plot_stuff"][0]), list(checkpoint["plot_stuff"][1])
        if checkpoint.get("amp_grad_scaler") and self.amp_grad_scaler:
            self.amp_grad_scaler.load_state_dict(checkpoint["amp_grad_scaler"])

    def reorder_UNet_components(self):
        """MH.py:1391-1408: re-register encoder -> decoder -> head; changes named_parameters() ORDER only."""
        net = self.network
        mods = {k: getattr(net, k) for k in ("conv_blocks_localization", "conv_blocks_context", "td", "tu", "seg_outputs")}
        for k in mods:
            delattr(net, k)
        for k in ("conv_blocks_context", "td", "tu", "conv_blocks_localization", "seg_outputs"):
            setattr(net, k, mods[k])
        net._named = list(net.named_parameters())
