"""``nnUNetTrainerLWF`` -- Learning without Forgetting.

Mirror of nnunet_ext/training/network_training/lwf/nnUNetTrainerLWF.py: ``initialize`` :96-108,
``run_training`` :124-296 (phase 1 head-only warm-up with frozen body and the plain loss :189-201, phase 2
teacher logits for every head :244-251, phase 3 LwF loss :253-261), ``run_iteration`` :298-370; and of
``calculate_target_logits`` (nnunet_ext/utilities/helpful_functions.py:207-266).

Reference behaviour kept in parity mode (pinned by tests/golden/trainer_reference.json:lwf_flow, which the reference's
own ``run_iteration`` / ``calculate_target_logits`` produced):
  * the distillation term enters the loss VALUE only -- predictions are detached (LWF.py:343), so it carries no
    gradient; targets are looked up by ``batch_idx % 250`` (:349);
  * ``tee(data_generator, 1)[0]`` (LWF.py:328,357) does not copy a generator, it ADVANCES it: head j is evaluated on
    batch k+j, the network trains on batch k+T, batch k+T+1 is dropped (:361) -- T+2 batches per iteration, and the KL
    compares predictions and teacher logits of unrelated patches.  ``same_batch_predictions=True`` is the fix behind a
    flag (what the reference's comments intend): every head is evaluated on the training batch, which lets the old heads'
    1x1x1 convs run on the body activations the training forward just produced (``engine.forward(body=False)``) instead
    of T extra full forwards;
  * the LwF branch also runs for the no-backprop iterations of the epoch loop (only ``freeze_run`` / ``do_val`` select the
    plain branch, LWF.py:303), so ``batch_idx`` advances there too;
  * the plain branch of the freeze run / validation is UPSTREAM's base iteration (``super(nnUNetTrainerV2, self)``,
    LWF.py:305): it does not clip the gradient norm at 12 (pinned by tests/golden/lwf_phase1_reference.*).
MI355X-first differences that do not change results:
  * teacher logits and predictions stay in HBM (288 GB) instead of round-tripping through host memory
    (``.cpu()`` at LWF.py:343, HF.py:254); the KL is one fused device reduction instead of CPU fp32 ops.
"""
import torch

from ....losses import DC_and_CE_loss, MultipleOutputLossLWF as LwFloss
from ..multihead.nnUNetTrainerMultiHead import nnUNetTrainerMultiHead

HYPERPARAMS = {'lwf_temperature': float}


def calculate_target_logits(mh_network, gen, num_batches_per_epoch, fp16=True, gpu_id=0):
    """HF.py:207-266: for each head IN TURN, ``num_batches_per_epoch`` consecutive batches of ``gen`` are
    pushed through body + that head (eval, identity nonlinearity) and the full-resolution logits are kept."""
    target_logits = dict()
    for task in list(mh_network.heads.keys()):
        target_logits[task] = list()
        for _ in range(num_batches_per_epoch):
            data_dict = next(gen)
            x = torch.as_tensor(data_dict['data'])
            target_logits[task].append(mh_network.head_logits(task, x))     # any --split_at (full forward for deeper splits)
    return target_logits


class nnUNetTrainerLWF(nnUNetTrainerMultiHead):
    def __init__(self, split, task, *args, lwf_temperature=2.0, same_batch_predictions=False, **kwargs):
        kwargs.setdefault("extension", "lwf")
        super().__init__(split, task, *args, **kwargs)
        self.lwf_temperature = lwf_temperature
        self.same_batch_predictions = same_batch_predictions
        self.freeze_run = True
        self.do_val = False
        self.batch_idx = 0
        self.target_logits = None

    def initialize(self, training=True, force_load_plans=False, num_epochs=500, prev_trainer_path=None,
                   call_for_eval=False):
        super().initialize(training, force_load_plans, num_epochs, prev_trainer_path, call_for_eval)
        self.loss_orig = self.loss
        loss_base = DC_and_CE_loss({'batch_dice': self.batch_dice, 'smooth': 1e-5, 'do_bg': False}, {})
        self.LwFloss = LwFloss(loss_base, self.ds_loss_weights, list(), list(), self.lwf_temperature)

    def run_training(self, task, output_folder=None, build_folder=True):
        if not self.was_initialized:
            self.initialize(True, num_epochs=self.max_num_epochs)
        if self.task != task or self.tr_gen is None:
            self.reinitialize(task)
        if str(task) not in self.mh_network.heads:
            self.mh_network.add_new_task(task, use_init=not self.transfer_heads)
        if len(self.mh_network.heads) == 1:
            # very first task: conventional training (LWF.py:173-184)
            self.freeze_run = False
            self.loss = self.loss_orig
            self.network = self.mh_network.assemble_model(task)
            ret = super().run_training(task, output_folder)
            self.freeze_run = True
            return ret
        # ---- phase 1: head-only training, body frozen, plain loss (LWF.py:189-201)
        self.freeze_run = True
        self.network = self.mh_network.assemble_model(task, freeze_body=True)
        self.loss = self.loss_orig
        self._run_epoch_loop()
        self.epoch = 0
        self.all_tr_losses, self.all_val_losses = [], []
        self.freeze_run = False
        # ---- phase 2: teacher logits of EVERY head with the pre-LwF body (LWF.py:236-251)
        self.network = self.mh_network.assemble_model(task, freeze_body=False)
        self.target_logits = calculate_target_logits(self.mh_network, self.tr_gen, self.num_batches_per_epoch, self.fp16)
        # ---- phase 3: train everything with the LwF loss (LWF.py:253-261)
        self.network.train()
        self.loss = self.LwFloss
        self.batch_idx = 0
        ret = super().run_training(task, output_folder)
        self.freeze_run = True
        return ret

    def _lwf_active(self):
        """LWF.py:303,309: the LwF branch runs unless freeze_run / do_val, and only with more than one head."""
        return not (self.freeze_run or self.do_val or len(self.mh_network.heads) <= 1 or self.loss is not self.LwFloss)

    def _targets(self, heads):
        return [self.target_logits[t][self.batch_idx % len(self.target_logits[t])] for t in heads[:-1]]

    def on_forward_done(self, data, output, do_backprop):
        """``same_batch_predictions`` mode: predictions of every head on the CURRENT batch, the old heads evaluated on
        the body activations the training forward just produced."""
        if not (self._lwf_active() and self.same_batch_predictions):
            return
        heads = list(self.mh_network.heads.keys())
        eng = self.network.engine_for(data)
        all_pred_logits = []
        with torch.no_grad():
            for t in heads:
                if t == str(self.mh_network.active_task):
                    all_pred_logits.append(output[0].detach())
                elif self.mh_network.head_is_seg_only(t):
                    all_pred_logits.append(eng.forward(data, seg_weights=self.mh_network.head_weights(t), body=False)[-1])
                else:
                    raise RuntimeError("same_batch_predictions re-uses the body pass of the training forward and needs "
                                       "split_at='seg_outputs'; other splits run the reference order (one forward per head)")
        self.loss.update_logits(all_pred_logits, self._targets(heads))

    def run_iteration(self, data_generator, do_backprop=True, run_online_evaluation=False, *args, **kwargs):
        if self.freeze_run or self.do_val:
            # LWF.py:303-308: ``super(nnUNetTrainerV2, self).run_iteration`` -- upstream's PLAIN iteration (zero_grad, forward,
            # loss, backward, optimizer step): no clip_grad_norm_(12), which lives in nnUNetTrainerV2.run_iteration that
            # this call skips -- followed by the head refresh.  Pinned by tests/golden/lwf_phase1_reference.* (the reference's
            # method executed with gradient norms of 50+).
            keep, self.max_grad_norm = self.max_grad_norm, None
            try:
                return super().run_iteration(data_generator, do_backprop, run_online_evaluation, *args, **kwargs)
            finally:
                self.max_grad_norm = keep
        if not self._lwf_active():
            return super().run_iteration(data_generator, do_backprop, run_online_evaluation, *args, **kwargs)
        if self.same_batch_predictions:
            ret = super().run_iteration(data_generator, do_backprop, run_online_evaluation, *args, **kwargs)
        else:
            # LWF.py:315-361 as the code runs: one batch PER HEAD for the predictions, the next one trains, one more is dropped
            heads = list(self.mh_network.heads.keys())
            preds = []
            with torch.no_grad():
                for t in heads:
                    x = torch.as_tensor(next(data_generator)['data']).to(self.device, non_blocking=True)
                    preds.append(self.mh_network.head_logits(t, x))
            self.loss.update_logits(preds, self._targets(heads))
            ret = super().run_iteration(data_generator, do_backprop, run_online_evaluation, *args, **kwargs)
            next(data_generator)
        self.batch_idx += 1       # LWF.py:364
        return ret

    def _perform_validation(self, *args, **kwargs):
        """LWF.py ``on_epoch_end``: the reference sets ``do_val = True`` around the per-head validation so that its
        iterations take the plain branch -- otherwise every validation iteration would consume T + 2 batches of the
        per-task generator, advance ``batch_idx`` (teacher logits out of step with training) and pair subject names with
        the wrong batch."""
        keep, self.do_val = self.do_val, True
        try:
            return super()._perform_validation(*args, **kwargs)
        finally:
            self.do_val = keep
