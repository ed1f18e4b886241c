"""``nnUNetTrainerRehearsalEWC`` -- BASELINE.json configs[4]: "Rehearsal + EWC mixed-task batches".

The reference has no single class for this (SURVEY.md 8a footnote): it is the rehearsal sampler
(rehearsal/nnUNetTrainerRehearsal.py:65-173) feeding the EWC trainer (ewc/nnUNetTrainerEWC.py:98-310).  The composite
keeps both behaviours untouched: training batches are drawn from the fused (current + sampled previous tasks) case list,
the loss is the EWC loss, ``after_train`` extracts Fisher / theta* on the fused generator exactly as the EWC trainer does
on its own.  Under data parallelism every rank draws its own mixed batches; gradients (and, in ``fisher_mode='accumulate'``,
the Fisher arena) are all-reduced as in the EWC trainer.
"""
from ..ewc.nnUNetTrainerEWC import nnUNetTrainerEWC
from ..rehearsal.nnUNetTrainerRehearsal import RehearsalMixin

HYPERPARAMS = {'ewc_lambda': float, 'samples_in_perc': float, 'seed': int}


class nnUNetTrainerRehearsalEWC(RehearsalMixin, nnUNetTrainerEWC):
    def __init__(self, split, task, *args, ewc_lambda=0.4, samples_in_perc=0.25, seed=3299, cases_per_task=40, **kwargs):
        kwargs.setdefault("extension", "rehearsal_ewc")
        super().__init__(split, task, *args, ewc_lambda=ewc_lambda, **kwargs)
        self._init_rehearsal(samples_in_perc, seed, cases_per_task)
