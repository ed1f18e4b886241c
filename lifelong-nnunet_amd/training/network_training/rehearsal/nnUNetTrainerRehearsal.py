"""``nnUNetTrainerRehearsal`` -- training set = current task + a seeded 25 % sample of every previous task.

Mirror of the sampling semantics of nnunet_ext/training/network_training/rehearsal/nnUNetTrainerRehearsal.py
:65-173 (``random.seed(self.seed)`` :73, per previous task in head order
``random.sample(dataset_tr.items(), round(len * samples))`` :132, validation = current task only :142,
``random.seed()`` reset :169).  The file-system part (loading each task's preprocessed folder) is out of scope;
a task's "dataset" here is its list of case identifiers and each batch draws cases uniformly from the fused
list, which yields the mixed-task batches of BASELINE config 5.
"""
import random

import numpy as np
import torch

from ....synthetic import make_patch_batch
from ..multihead.nnUNetTrainerMultiHead import nnUNetTrainerMultiHead

HYPERPARAMS = {'samples_in_perc': float, 'seed': int}


def task_cases(task, n_cases=40):
    """Synthetic stand-in for ``dataset_tr`` keys (sorted, as MH.py:273-277 produces them)."""
    return [f"{task}_{i:03d}" for i in range(1, n_cases + 1)]


class RehearsalPatchGenerator:
    """Draws ``batch_size`` cases uniformly (numpy RNG, like upstream DataLoader3D) from the fused case list;
    every case maps to a deterministic synthetic patch."""

    def __init__(self, cases, plans, seed=12345):
        self.cases, self.plans = list(cases), plans
        self.rng = np.random.RandomState(seed)
        self._cache = {}

    def _patch(self, case):
        if case not in self._cache:
            h = sum(ord(c) * (i + 1) for i, c in enumerate(case)) % 100003
            d, t = make_patch_batch(1, self.plans["patch_size"], self.plans["num_pool"], self.plans["num_input_channels"],
                                    self.plans["num_classes"], seed=h)
            self._cache[case] = (d, t)
        return self._cache[case]

    def __iter__(self):
        return self

    def __next__(self):
        keys = [self.cases[i] for i in self.rng.choice(len(self.cases), self.plans["batch_size"], True)]
        ds, ts = zip(*[self._patch(k) for k in keys])
        data = torch.cat(ds, 0)
        target = [torch.cat([t[i] for t in ts], 0) for i in range(len(ts[0]))]
        return {'data': data, 'target': target, 'keys': keys}


class nnUNetTrainerRehearsal(nnUNetTrainerMultiHead):
    def __init__(self, split, task, *args, samples_in_perc=0.25, seed=3299, cases_per_task=40, **kwargs):
        kwargs.setdefault("extension", "rehearsal")
        super().__init__(split, task, *args, **kwargs)
        assert 0 < samples_in_perc <= 1, "Your provided samples are not in the correct range (0, 1]"
        self.samples, self.seed, self.cases_per_task = samples_in_perc, seed, cases_per_task
        self.dataset_tr = None

    def get_basic_generators(self, use_all_data=False):
        random.seed(self.seed)                                               # REH.py:73
        dataset_tr_fused = list(task_cases(self.task, self.cases_per_task))
        try:
            tasks_in_head = [t for t in self.mh_network.heads.keys() if t != str(self.task)]
        except AttributeError:
            tasks_in_head = []
        self.sampled = {}
        for task in tasks_in_head:                                            # head order (REH.py:107)
            items = task_cases(task, self.cases_per_task)
            sample_tr = random.sample(items, round(len(items) * self.samples))   # REH.py:132
            self.sampled[task] = sample_tr
            dataset_tr_fused += sample_tr
        self.dataset_tr = dataset_tr_fused
        random.seed()                                                         # REH.py:169
        dl_tr = RehearsalPatchGenerator(dataset_tr_fused, self.plans, seed=12345 + self.fold)
        dl_val = self.data_provider(self.task, "val", self.plans)             # validation: current task only (REH.py:142)
        return dl_tr, dl_val

    def reinitialize(self, task, print_loss_info=True):
        self.task = task
        self.tr_gen, self.val_gen = self.get_basic_generators()

    def initialize(self, training=True, force_load_plans=False, num_epochs=500, prev_trainer_path=None,
                   call_for_eval=False):
        super().initialize(training, force_load_plans, num_epochs, prev_trainer_path, call_for_eval)
        if training:
            self.tr_gen, self.val_gen = self.get_basic_generators()
