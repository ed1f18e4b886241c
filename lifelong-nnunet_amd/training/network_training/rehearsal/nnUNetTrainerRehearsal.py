"""``nnUNetTrainerRehearsal`` -- training set = current task + a seeded 25 % sample of every previous task.

Mirror of nnunet_ext/training/network_training/rehearsal/nnUNetTrainerRehearsal.py:65-173
(``get_basic_generators``), pinned by ``tests/golden/trainer_reference.json:rehearsal`` which the REFERENCE's own
method produced (oracle/make_goldens_trainers.py):
  * ``random.seed(self.seed)`` (REH.py:73); the current task's dataset is loaded and split (REH.py:76-77);
  * for every task already in ``mh_network.heads`` -- in head order, the current task is NOT excluded by the reference
    loop, it simply is not a head yet when the generators are built -- that task's dataset is loaded, split with the SAME
    fold, and ``random.sample(dataset_tr.items(), round(len(dataset_tr) * samples))`` (REH.py:127-132) is merged into the
    fused training dictionary (current task's training cases first, then the samples in draw order);
  * validation uses the current task only (REH.py:142); ``random.seed()`` afterwards (REH.py:169).
The fused dictionary feeds ONE loader that draws cases uniformly (upstream ``DataLoader3D``), which yields the mixed-task
batches of BASELINE config 5.

Where the cases come from is the ``data_provider``'s business: a provider that offers ``dataset_for(task)`` /
``splits_file_for(task)`` / ``generator_for(dataset, plans, split)`` (``dataloading.PreprocessedDataProvider``: real
nnU-Net-preprocessed folders) is used for BOTH splits; the default synthetic provider falls back to deterministic
synthetic patches per case identifier.  ``RehearsalMixin`` carries the behaviour so that it can be combined with another
trainer (``rehearsal_ewc/nnUNetTrainerRehearsalEWC.py`` = BASELINE config 5).
"""
import random
from collections import OrderedDict

import numpy as np
import torch

from ....dataloading import do_split
from ....synthetic import make_patch_batch
from ..multihead.nnUNetTrainerMultiHead import nnUNetTrainerMultiHead

HYPERPARAMS = {'samples_in_perc': float, 'seed': int}


def task_cases(task, n_cases=40):
    """Synthetic stand-in for a task's case identifiers (sorted, as ``load_dataset`` returns them)."""
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
                                    self.plans["num_classes"], seed=h, pool_op_kernel_sizes=self.plans.get("pool_op_kernel_sizes"))
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


class RehearsalMixin:
    """``get_basic_generators`` / ``reinitialize`` / ``initialize`` of the rehearsal trainer, for any multi-head trainer."""

    def _init_rehearsal(self, samples_in_perc, seed, cases_per_task):
        assert 0 < samples_in_perc <= 1, "Your provided samples are not in the correct range (0, 1]"
        self.samples, self.seed, self.cases_per_task = samples_in_perc, seed, cases_per_task
        self.dataset = self.dataset_tr = self.dataset_val = None
        self.sampled = {}

    # ---- where a task's cases come from
    def _task_dataset(self, task):
        dp = self.data_provider
        if hasattr(dp, "dataset_for"):
            return dp.dataset_for(task)
        return OrderedDict((k, {"case": k}) for k in task_cases(task, self.cases_per_task))

    def _split(self, dataset, task):
        dp = self.data_provider
        return do_split(dataset, self.fold, dp.splits_file_for(task) if hasattr(dp, "splits_file_for") else None)

    def _generator(self, dataset, split):
        dp = self.data_provider
        if hasattr(dp, "generator_for"):
            return dp.generator_for(dataset, self.plans, split)
        if split == "val":
            return dp(self.task, "val", self.plans)
        return RehearsalPatchGenerator(list(dataset.keys()), self.plans, seed=12345 + self.fold)

    def get_basic_generators(self, use_all_data=False):
        random.seed(self.seed)                                               # REH.py:73
        self.dataset = self._task_dataset(self.task)                         # REH.py:76-77
        self.dataset_tr, self.dataset_val = self._split(self.dataset, self.task)
        dataset_fused, dataset_tr_fused = OrderedDict(self.dataset), OrderedDict(self.dataset_tr)
        try:
            tasks_in_head = list(self.mh_network.heads.keys())               # REH.py:88-91
        except AttributeError:
            tasks_in_head = []
        self.sampled = {}
        for task in tasks_in_head:                                            # head order (REH.py:107)
            ds = self._task_dataset(task)                                     # REH.py:127-128
            ds_tr, _ = self._split(ds, task)
            items = list(ds_tr.items())
            sample_tr = random.sample(items, round(len(ds_tr) * self.samples))    # REH.py:132
            self.sampled[task] = [k for k, _ in sample_tr]
            dataset_fused.update(ds)                                          # REH.py:135-136
            dataset_tr_fused.update(sample_tr)
        self.dataset, self.dataset_tr = dataset_fused, dataset_tr_fused      # REH.py:140-142: validation stays the current task's
        random.seed()                                                         # REH.py:169
        return self._generator(self.dataset_tr, "train"), self._generator(self.dataset_val, "val")

    def reinitialize(self, task, print_loss_info=True):
        super().reinitialize(task, print_loss_info)
        self.tr_gen, self.val_gen = self.get_basic_generators()

    def initialize(self, training=True, force_load_plans=False, num_epochs=500, prev_trainer_path=None,
                   call_for_eval=False):
        super().initialize(training, force_load_plans, num_epochs, prev_trainer_path, call_for_eval)
        if training:
            self.tr_gen, self.val_gen = self.get_basic_generators()


class nnUNetTrainerRehearsal(RehearsalMixin, nnUNetTrainerMultiHead):
    def __init__(self, split, task, *args, samples_in_perc=0.25, seed=3299, cases_per_task=40, **kwargs):
        kwargs.setdefault("extension", "rehearsal")
        super().__init__(split, task, *args, **kwargs)
        self._init_rehearsal(samples_in_perc, seed, cases_per_task)
