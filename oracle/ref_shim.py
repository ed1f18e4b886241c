"""Import shim that lets the REFERENCE's trainer modules (``/root/reference/nnunet_ext/training/network_training/*``)
be imported and their methods executed verbatim in the build container, where the un-vendored upstream packages
(``nnunet`` @77bc485, ``batchgenerators`` 0.21, requirements.txt:3-4) are not installed.

TEST INFRASTRUCTURE ONLY (used by ``oracle/make_goldens_trainers.py`` to generate ``tests/golden/trainer_reference.*``;
never imported by the product package, never shipped to the GPU box as a dependency of anything that runs there).

What is stubbed, and how:
  * every ``nnunet.*`` / ``batchgenerators.*`` (and a few plotting / imaging packages the reference imports at module
    level) module resolves to an empty stand-in whose attributes are inert dummy classes -- enough for ``import`` and
    ``class X(nnUNetTrainerV2)`` statements to succeed;
  * the handful of upstream symbols the executed METHODS really call get small concrete definitions below, each a
    restatement of upstream's published behaviour: ``maybe_to_torch`` / ``to_cuda``,
    ``softmax_helper``, ``sum_tensor``, ``MultipleOutputLoss2``, ``RobustCrossEntropyLoss``, ``DC_and_CE_loss`` (= the
    oracle's Dice+CE), and the file helpers of ``batchgenerators.utilities.file_and_folder_operations``;
  * ``cuda_as_cpu()`` lets lines such as ``torch.tensor([1], device='cuda:0')`` (EWC.py:301) or ``x.to(0)`` (RW.py:246)
    run on the CPU by mapping CUDA device arguments to the CPU for the duration of a call.
Nothing of the reference (source or bytecode) is copied: the reference code is imported from where it lies.
"""
from __future__ import annotations

import contextlib
import importlib.abc
import importlib.machinery
import json
import os
import pickle
import sys
import types

import numpy as np
import torch
from torch import nn

REF = "/root/reference"
_STUB_ROOTS = ("nnunet", "batchgenerators", "timm", "SimpleITK", "medpy", "skimage", "nibabel", "seaborn", "matplotlib")


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Dummy()


class _StubModule(types.ModuleType):
    __all__: list = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (_Dummy,), {})
        setattr(self, name, cls)
        return cls


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def _concrete(name, **attrs):
    m = _StubModule(name)
    m.__path__ = []
    m.__dict__.update(attrs)
    m.__all__ = list(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


class MultipleOutputLoss2(nn.Module):
    """upstream deep_supervision.MultipleOutputLoss2: zero-weight levels are skipped."""

    def __init__(self, loss, weight_factors=None):
        super().__init__()
        self.weight_factors, self.loss = weight_factors, loss

    def forward(self, x, y):
        w = self.weight_factors if self.weight_factors is not None else [1] * len(x)
        l = w[0] * self.loss(x[0], y[0])
        for i in range(1, len(x)):
            if w[i] != 0:
                l += w[i] * self.loss(x[i], y[i])
        return l


class RobustCrossEntropyLoss(nn.CrossEntropyLoss):
    def forward(self, input, target):
        if len(target.shape) == len(input.shape):
            target = target[:, 0]
        return super().forward(input, target.long())


class DC_and_CE_loss(nn.Module):
    """upstream dice_loss.DC_and_CE_loss(soft_dice_kwargs, ce_kwargs) -> CE + soft Dice, via the oracle."""

    def __init__(self, soft_dice_kwargs, ce_kwargs, aggregate="sum", square_dice=False, weight_ce=1, weight_dice=1,
                 log_dice=False, ignore_label=None):
        super().__init__()
        assert soft_dice_kwargs.get("smooth", 1e-5) == 1e-5 and not soft_dice_kwargs.get("do_bg", False)
        self.batch_dice = soft_dice_kwargs.get("batch_dice", False)

    def forward(self, net_output, target):
        from . import losses
        return losses.dc_and_ce_loss(net_output, target, self.batch_dice)


def sum_tensor(inp, axes, keepdim=False):
    """upstream tensor_utilities.sum_tensor: sum over the given axes (largest first)."""
    axes = np.unique(axes).astype(int)
    if keepdim:
        for ax in axes:
            inp = inp.sum(int(ax), keepdim=True)
    else:
        for ax in sorted(axes, reverse=True):
            inp = inp.sum(int(ax))
    return inp


def maybe_to_torch(d):
    if isinstance(d, list):
        return [maybe_to_torch(i) if not isinstance(i, torch.Tensor) else i for i in d]
    if not isinstance(d, torch.Tensor):
        return torch.from_numpy(d).float()
    return d


class nnUNetTrainer:
    """Stand-in for upstream ``nnUNetTrainer`` / ``NetworkTrainer`` (nnunet @ 77bc485, not in the reference tree): the ONE
    upstream method the reference reaches around its own overrides -- ``super(nnUNetTrainerV2, self).run_iteration(...)`` in
    the LwF freeze run (lwf/nnUNetTrainerLWF.py:303-305) lands on ``NetworkTrainer.run_iteration``, restated here from its
    published form: zero_grad, forward, loss, backward, optimizer step -- NO gradient clipping (the clip at 12 lives in
    ``nnUNetTrainerV2.run_iteration``, which that ``super`` call skips) and no head bookkeeping."""

    def run_iteration(self, data_generator, do_backprop=True, run_online_evaluation=False):
        data_dict = next(data_generator)
        data = maybe_to_torch(data_dict['data'])
        target = maybe_to_torch(data_dict['target'])
        self.optimizer.zero_grad()
        assert not self.fp16, "the fixtures run the reference's fp32 branch on the CPU"
        output = self.network(data)
        del data
        l = self.loss(output, target)
        if do_backprop:
            l.backward()
            self.optimizer.step()
        if run_online_evaluation:
            self.run_online_evaluation(output, target)
        del target
        return l.detach().cpu().numpy()


class nnUNetTrainerV2(nnUNetTrainer):
    """Inert base class: the reference's trainers only need it to exist; the methods we execute are the reference's own."""

    def __init__(self, *a, **k):
        pass

    def print_to_log_file(self, *a, **k):
        pass


_installed = False


def install():
    """Idempotent: register the stub finder + the concrete upstream stand-ins, put /root/reference on sys.path."""
    global _installed
    if _installed:
        return
    _installed = True
    sys.meta_path.insert(0, _StubFinder())
    for quiet in ("nnUNet_raw_data_base", "nnUNet_preprocessed", "RESULTS_FOLDER", "EVALUATION_FOLDER", "PARAM_SEARCH_FOLDER"):
        os.environ.setdefault(quiet, "/tmp/lnn_ref_shim/" + quiet)
    import importlib
    for pkg in ("nnunet", "nnunet.utilities", "nnunet.training", "nnunet.training.loss_functions",
                "nnunet.training.network_training", "batchgenerators", "batchgenerators.utilities"):
        importlib.import_module(pkg)
    _concrete("nnunet.utilities.to_torch", maybe_to_torch=maybe_to_torch, to_cuda=lambda x, non_blocking=True, gpu_id=0: x)
    _concrete("nnunet.utilities.nd_softmax", softmax_helper=lambda x: torch.softmax(x, 1))
    _concrete("nnunet.utilities.tensor_utilities", sum_tensor=sum_tensor)
    _concrete("nnunet.training.loss_functions.deep_supervision", MultipleOutputLoss2=MultipleOutputLoss2)
    _concrete("nnunet.training.loss_functions.crossentropy", RobustCrossEntropyLoss=RobustCrossEntropyLoss)
    _concrete("nnunet.training.loss_functions.dice_loss", DC_and_CE_loss=DC_and_CE_loss)
    _concrete("nnunet.training.network_training.nnUNetTrainerV2", nnUNetTrainerV2=nnUNetTrainerV2)

    def _load_pickle(f, mode="rb"):
        with open(f, mode) as fh:
            return pickle.load(fh)

    def _write_pickle(obj, f, mode="wb"):
        with open(f, mode) as fh:
            pickle.dump(obj, fh)

    def _save_json(obj, f, indent=4, sort_keys=True):
        with open(f, "w") as fh:
            json.dump(obj, fh, sort_keys=sort_keys, indent=indent)

    def _load_json(f):
        with open(f) as fh:
            return json.load(fh)

    _concrete("batchgenerators.utilities.file_and_folder_operations", join=os.path.join, isfile=os.path.isfile,
              isdir=os.path.isdir, os=os, maybe_mkdir_p=lambda d: os.makedirs(d, exist_ok=True), load_pickle=_load_pickle,
              write_pickle=_write_pickle, save_pickle=_write_pickle, save_json=_save_json, load_json=_load_json,
              subfiles=lambda d, join=True, prefix=None, suffix=None, sort=True: sorted(
                  (os.path.join(d, f) if join else f) for f in os.listdir(d)
                  if os.path.isfile(os.path.join(d, f)) and (prefix is None or f.startswith(prefix))
                  and (suffix is None or f.endswith(suffix))),
              subdirs=lambda d, join=True, prefix=None, suffix=None, sort=True: sorted(
                  (os.path.join(d, f) if join else f) for f in os.listdir(d) if os.path.isdir(os.path.join(d, f))))
    if REF not in sys.path:
        sys.path.insert(0, REF)


def _is_cuda_dev(d):
    if isinstance(d, int) and not isinstance(d, bool):
        return True
    if isinstance(d, str):
        return d.startswith("cuda")
    if isinstance(d, torch.device):
        return d.type == "cuda"
    return False


@contextlib.contextmanager
def cuda_as_cpu():
    """Run reference lines that name CUDA devices explicitly on the CPU (values are device independent)."""
    orig_to, orig_tensor, orig_zeros_like, orig_cuda = torch.Tensor.to, torch.tensor, torch.zeros_like, torch.Tensor.cuda

    def to(self, *a, **k):
        a = tuple(x for x in a if not _is_cuda_dev(x))
        if _is_cuda_dev(k.get("device")):
            k.pop("device")
        if not a and not k:
            return self
        return orig_to(self, *a, **k)

    def _strip(fn):
        def wrapped(*a, **k):
            if _is_cuda_dev(k.get("device")):
                k.pop("device")
            return fn(*a, **k)
        return wrapped

    torch.Tensor.to = to
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.tensor = _strip(orig_tensor)
    torch.zeros_like = _strip(orig_zeros_like)
    try:
        yield
    finally:
        torch.Tensor.to, torch.tensor, torch.zeros_like, torch.Tensor.cuda = orig_to, orig_tensor, orig_zeros_like, orig_cuda


def import_trainer(ext, cls):
    install()
    import importlib
    mod = importlib.import_module(f"nnunet_ext.training.network_training.{ext}.{cls}")
    return mod, getattr(mod, cls)
