"""``MultiHead_Module`` with the reference's interface and state-dict naming on top of the flat arena.

Mirrors nnunet_ext/network_architecture/MultiHead_Module.py:
  constructor / split path handling :16-125, ``forward`` :127-137, ``update_after_iteration`` :139-157,
  ``assemble_model`` :326-377, ``_set_requires_grad`` :379-395, ``add_new_task`` :435-458,
  ``add_n_tasks_and_activate`` :460-485, getters / setters / ``replace_layers`` :488-572.
Semantics kept: body parameters are SHARED tensors with the running model, head parameters are
per-task copies; the head of the active task is refreshed from the running model after every training
iteration.  What is dropped is the cost: the reference re-splits the module tree and ``deepcopy``s the
head every iteration (MHM.py:153-157,324) and deep-copies + ``load_state_dict``s on every
``assemble_model`` (MHM.py:343-359); here both are a handful of device-to-device copies of the head
tensors (2 400 floats for the 5-level U-Net).  NESTED splits follow the reference after construction too (``_reference_resplit``).
"""
from __future__ import annotations

import copy
from collections import OrderedDict
from typing import Type

import warnings

import torch
from torch import nn


_MODULE_FIELDS = set(nn.Module().__dict__.keys())


def _shell_like(ref: nn.Module) -> nn.Module:
    """An EMPTY module of ``ref``'s class (no parameters, no children) carrying its plain attributes, plus copies of its
    parameter-free leaf children (``lrelu`` of a ``ConvDropoutNormNonlin``): the bodies and heads the reference splits off are module
    trees (MHM.py:159-324) -- ``heads[task].children()`` yields ``seg_outputs`` as the ModuleList it is in the model, a decoder block as a
    ``StackedConvLayers`` -- and code written against them walks those trees (``replace_layers``, MHM.py:544-572)."""
    new = ref.__class__.__new__(ref.__class__)
    nn.Module.__init__(new)
    for k, v in ref.__dict__.items():
        if k not in _MODULE_FIELDS:
            new.__dict__[k] = copy.copy(v)
    for name, child in ref.named_children():
        if next(child.parameters(), None) is None and next(child.children(), None) is None:
            setattr(new, name, copy.deepcopy(child))
    return new


def _set_nested(root: nn.Module, dotted: str, param: nn.Parameter, like: nn.Module = None):
    """Register ``param`` under its dotted name below ``root``, creating the containers on the way: typed like the modules at the same
    paths of ``like`` (the network) when given, plain ``nn.Module`` holders otherwise."""
    parts = dotted.split(".")
    m, ref = root, like
    for p in parts[:-1]:
        ref = getattr(ref, p, None) if ref is not None else None
        if not hasattr(m, p):
            setattr(m, p, _shell_like(ref) if isinstance(ref, nn.Module) else nn.Module())
        m = getattr(m, p)
    m.register_parameter(parts[-1], param)


class MultiHead_Module(nn.Module):
    # What a NESTED split ('tu.1', ...) does after construction: True = the reference's behaviour (see _reference_resplit), False = the
    # construction-time partition is kept.  Top-level splits ('seg_outputs', 'tu', ...) are not affected.
    reference_nested_resplit = True
    _warned_nested_resplit = False

    def __init__(self, class_object: Type[nn.Module], split_at, task, prev_trainer=None, *args, **kwargs):
        super().__init__()
        self.class_object = class_object
        if prev_trainer is None:
            self.model = class_object(*args, **kwargs)
        else:
            assert isinstance(prev_trainer, class_object), \
                "This function splits a '{}' module class object, but a '{}' module is provided.".format(
                    class_object.__name__, type(prev_trainer))
            assert len(list(prev_trainer.children())) > 0, \
                "When using a prev_trainer, please ensure that it is not empty or do not specify one."
            self.model = prev_trainer
        assert isinstance(split_at, str), "The provided split needs to be a string.."
        self.split = [x.strip() for x in split_at.split('.')]
        names = [n for n, _ in self.model.named_parameters()]
        self._check_and_simplify_split()
        self.heads = nn.ModuleDict()
        assert isinstance(task, (str, int)), "The provided task needs to be an integer (ID) or string, not {}..".format(type(task))
        self.active_task = task

        # everything at or after the split MODULE in registration (pre-order) order is the head; the position is taken from the
        # module tree, so a split at a parameter-less module (``td`` with convolutional pooling: the reference accepts it) works
        path, first, seen = '.'.join(self.split), None, 0
        for mod_name, mod in self.model.named_modules():
            if mod_name == path:
                first = seen
                break
            seen += len(mod._parameters)
        assert first is not None, "The provided split path '{}' does not exist..".format(path)
        assert first > 0, "You tried to split before the first layer, so the body would be empty --> body can never be empty.."
        self._head_names = names[first:]
        # Reference behaviour, reproduced (tests/golden/multihead_splits_reference.json, produced by the reference class): for a
        # NESTED split ('tu.1', 'conv_blocks_context.2', ...) the body keeps the WHOLE top-level container the split lies in --
        # the container object is shared with the running model, so MHM.py:254-262 cannot take the head-side members out of it.
        # They are then both body (shared tensors, ``body.*`` state-dict keys, frozen by ``freeze_body``) and head (per-task
        # copies that ``assemble_model`` writes back).  Top-level splits ('seg_outputs', 'tu') have no such overlap.
        top = self.split[0] + '.'
        self._body_names = [n for i, n in enumerate(names) if i < first or (len(self.split) > 1 and n.startswith(top))]
        self._resplit = False
        self._share_body()
        init_module = self._head_from_model()
        self.state_init = OrderedDict((k, v.clone()) for k, v in init_module.state_dict().items())
        self.heads[str(task)] = init_module
        self.body_freezed = True
        self.assemble_model(task, freeze_body=False)
        self.body_freezed = False

    # ------------------------------------------------------------------------------------------ split
    def _check_and_simplify_split(self):
        from operator import attrgetter

        def module_at(path):
            try:
                m = attrgetter('.'.join(path))(self.model)
            except AttributeError:
                return None
            return m if isinstance(m, nn.Module) else None
        assert len(self.split) > 0 and self.split != [''] and module_at(self.split) is not None, \
            "The provided split path '{}' does not exist..".format('.'.join(self.split))
        # MHM.py:73-92: drop trailing components that name the FIRST child of their parent
        full = self.split[:]
        while len(self.split) > 1:
            first_child = next(module_at(self.split[:-1]).named_children())[0]
            if self.split[-1] == first_child:
                self.split = self.split[:-1]
            else:
                break
        first_top = next(self.model.named_children())[0]
        assert not (len(self.split) == 1 and self.split[0] == first_top), \
            "The provided split '{}' is empty after simplification and would split before the first layer..".format('.'.join(full))

    def get_split_path(self):
        return '.'.join(self.split)

    def _share_body(self):
        """``self.body`` = the body parameters of the running model: the SAME Parameter objects (MHM.py:108: the split hands the
        model's own modules to the body)."""
        params = dict(self.model.named_parameters())
        self.body = nn.Module()
        for n in self._body_names:
            _set_nested(self.body, n, params[n], self.model)
        self._body_detached = False

    def _reference_resplit(self):
        """NESTED splits only -- the partition the reference is in from its FIRST re-split on (``update_after_iteration``,
        MHM.py:139-157), reproduced tensor by tensor (tests/golden/multihead_nested_flow_reference.json, produced by executing the
        reference class).  Its recursive split keeps its working objects in mutable default arguments (MHM.py:159-160): the second
        and every later call starts with the path list the constructor's call left behind, never descends into the split
        container, and returns
          * a body that holds EVERY top-level module: all tensors are body from here on (shared with the running model, ``body.*``
            state-dict keys, frozen by ``freeze_body``), the segmentation layers included;
          * a copy of the default-argument head, which holds the members of the INNERMOST split container from the split index on
            (``tu.1`` -> ``tu.1..``; ``conv_blocks_context.1.blocks.1`` -> ``conv_blocks_context.1.blocks.1..``) as the module
            objects the constructor took out of the model -- they left the running model at the first ``assemble_model`` and keep
            their CONSTRUCTION-TIME values (``state_init``'s) whatever the training does.
        So the active task's head is replaced by those few tensors at their initial values on every update, ``assemble_model`` writes
        them back over the trained ones, and ``add_new_task(use_init=True)`` raises (``state_init`` has the construction-time keys).
        ``MultiHead_Module.reference_nested_resplit = False`` keeps the construction-time partition instead (every tensor from the
        split on is per-task and is refreshed from the running model -- what the reference's documentation describes)."""
        if not MultiHead_Module._warned_nested_resplit:
            MultiHead_Module._warned_nested_resplit = True
            warnings.warn(
                f"MultiHead_Module(split_at={'.'.join(self.split)!r}): reproducing the reference's nested re-split (MHM.py:159-160, mutable "
                "default arguments): from now on every tensor is body, the active head is reset to its construction-time values on each "
                "update_after_iteration, and add_new_task(use_init=True) raises.  Set MultiHead_Module.reference_nested_resplit = False "
                "for the documented semantics (per-task tensors from the split on).", stacklevel=3)
        parent = '.'.join(self.split[:-1]) + '.'
        self._head_names = [n for n in self._head_names if n.startswith(parent)]
        self._body_names = [n for n, _ in self.model.named_parameters()]
        if not self._body_detached:
            self._share_body()
        self._resplit = True

    def _head_from_model(self, model=None):
        params = dict((self.model if model is None else model).named_parameters())
        head = nn.Module()
        for n in self._head_names:
            _set_nested(head, n, nn.Parameter(params[n].detach().clone(), requires_grad=params[n].requires_grad), self.model)
        return head

    # ------------------------------------------------------------------------------------------ API
    def forward(self, x):
        return self.class_object.forward(self.model, x)

    def update_after_iteration(self, model=None, update_body=True):
        """Refresh the active task's head from the running model (body tensors are shared already)."""
        model = self.model if model is None else model
        if len(self.split) > 1 and not self._resplit and self.reference_nested_resplit:
            self._reference_resplit()
        if update_body and self._body_detached:
            self._share_body()          # MHM.py:152-153: the body is re-split from the running model, a body given to set_body is dropped
        head = self.heads[str(self.active_task)]
        if self._resplit:               # see _reference_resplit: the head the reference's re-split returns holds construction-time values
            if [n for n, _ in head.named_parameters()] != self._head_names:
                head = nn.Module()
                for n in self._head_names:
                    _set_nested(head, n, nn.Parameter(self.state_init[n].clone()), self.model)
                self.heads[str(self.active_task)] = head
            else:
                with torch.no_grad():
                    for n, p in head.named_parameters():
                        p.copy_(self.state_init[n])
            return
        src = dict(model.named_parameters())
        with torch.no_grad():
            for n, p in head.named_parameters():
                p.copy_(src[n])

    def assemble_model(self, task, freeze_body=False):
        if self.active_task == task and freeze_body == self.body_freezed:
            return self.model
        assert str(task) in self.heads.keys(), \
            "The provided task '{}' is not a known head, so either initialize the task or provide one that already exists: {}.".format(
                task, list(self.heads.keys()))
        dst = dict(self.model.named_parameters())
        with torch.no_grad():
            if self._body_detached:     # a body given to set_body reaches the running model here (MHM.py:343-359), then is shared again
                for n, p in self.body.named_parameters():
                    if n in dst:
                        dst[n].copy_(p)
                self._share_body()
            for n, p in self.heads[str(task)].named_parameters():
                dst[n].copy_(p)
        if hasattr(self.model, "mark_params_changed"):
            self.model.mark_params_changed()
        self.active_task = task
        if freeze_body and not self.body_freezed:
            self._set_requires_grad(False)
            self.body_freezed = True
        if not freeze_body and self.body_freezed:
            self._set_requires_grad(True)
            self.body_freezed = False
        return self.model

    def _set_requires_grad(self, requires_grad):
        body = set(self._body_names)
        for name, param in self.model.named_parameters():
            if name in body:
                param.requires_grad = requires_grad

    def add_new_task(self, task, use_init, model=None):
        if model is None:
            last = self.heads[list(self.heads.keys())[-1]]
            new = nn.Module()
            for n, p in last.named_parameters():
                _set_nested(new, n, nn.Parameter(p.detach().clone()), self.model)
            self.heads[str(task)] = new
            if use_init:        # MHM.py:448-452: registered first, so a refused state_init (nested split after a re-split) leaves it behind
                new.load_state_dict(self.state_init)
        else:
            new = nn.Module()
            for n, p in model.named_parameters():
                _set_nested(new, n, nn.Parameter(p.detach().clone()), self.model)
            self.heads[str(task)] = new

    def add_n_tasks_and_activate(self, list_of_tasks, activate_with, remove_old_tasks=True):
        for task in list_of_tasks:
            if str(task) not in self.heads:
                self.add_new_task(task, use_init=True)
        if remove_old_tasks:
            for task in list(self.heads.keys()):
                if task not in [str(t) for t in list_of_tasks]:
                    del self.heads[task]
        self.assemble_model(activate_with)

    # ------------------------------------------------------------------------------------------ getters / setters (MHM.py:488-572)
    def get_heads(self):
        """A deep copy of the ModuleDict of heads (MHM.py:488-493)."""
        return copy.deepcopy(self.heads)

    def get_body(self):
        """A deep copy of the body (MHM.py:495-500): independent tensors, not the running model's."""
        return copy.deepcopy(self.body)

    def set_heads(self, heads, reset=True):
        """Replace (``reset=True``) or update the ModuleDict of heads (MHM.py:502-517); the running model is untouched until the next
        ``assemble_model``."""
        assert isinstance(heads, nn.ModuleDict), "Provided heads are not a nn.ModuleDict."
        if reset:
            del self.heads
            self.heads = heads
        else:
            self.heads.update(heads)

    def set_body(self, body):
        """Replace the body by a COPY of ``body`` (MHM.py:519-528).  As in the reference the running model only takes these values at
        the next ``assemble_model`` that does not return early; ``update_after_iteration(update_body=True)`` re-derives the body from
        the running model and drops them."""
        assert isinstance(body, nn.Module), "Provided body is not a nn.Module.."
        del self.body
        self.body = copy.deepcopy(body)
        self._body_detached = True

    def replace_layers(self, model, old, new):
        """Every child of ``model`` (recursively) that is an instance of ``old`` is replaced by ``new`` (MHM.py:544-572).  The body and
        the heads of THIS class are module trees of the network's own classes (``StackedConvLayers`` / ``ConvDropoutNormNonlin`` /
        ``nn.Sequential`` / ``nn.ModuleList`` / ``nn.LeakyReLU``; the leaves that hold ``weight`` / ``bias`` are parameter holders, the
        convolutions themselves run in the HIP engine), so the walk finds them as it finds the reference's -- e.g. every ``nn.LeakyReLU``
        of a head -- but swapping a parameter holder for an ``nn.Conv3d`` does not change what the engine computes."""
        assert model is not None and new is not None and old is not None, \
            "To replace a Module, the layers need to be Modules as well as the model.."
        for name, module in model.named_children():
            if len(list(module.children())) > 0:
                self.replace_layers(module, old, new)
            if isinstance(module, old):
                setattr(model, name, new)
        return model

    def head_weights(self, task):
        """seg-head tensors of ``task`` in engine order (used for multi-head evaluation on one body pass).  Only valid when
        the head IS the segmentation layers (split ``seg_outputs``, the split the reference's tests and docs use,
        test_multi_head_trainer.py:149): for a deeper split the head also holds decoder blocks, which a different set of
        1x1x1 weights on shared body activations cannot express -- refuse instead of silently mis-reading tensors."""
        named = list(self.heads[str(task)].named_parameters())
        assert all(n.startswith("seg_outputs.") for n, _ in named), \
            "multi-head evaluation on one body pass needs split_at='seg_outputs' (head of task '{}' holds {}); " \
            "assemble_model(task) + a full forward per head is required for deeper splits".format(task, [n for n, _ in named][:3])
        return [p for _, p in sorted(named, key=lambda kv: int(kv[0].split('.')[1]))]

    def head_is_seg_only(self, task=None):
        """True when the head holds nothing but the 1x1x1 segmentation layers (split ``seg_outputs``)."""
        task = self.active_task if task is None else task
        return all(n.startswith("seg_outputs.") for n, _ in self.heads[str(task)].named_parameters())

    def head_logits(self, task, x):
        """Full-resolution logits of head ``task`` on ``x`` (eval, identity nonlinearity, no autograd) -- what the reference
        computes with ``assemble_model(task)`` + a complete forward per head (MHM.py:326-377, LWF.py:317-346, HF.py:239-258).
        For the ``seg_outputs`` split that is the task's 1x1x1 weights on one body pass; for any other ``--split_at``
        (run_training.py:103) the head holds body-side layers too, so the head is swapped in, the whole network runs, and
        the previously active head (and the body's frozen state) are restored."""
        if self.head_is_seg_only(task):
            return self.model.forward_heads(x, [self.head_weights(task)])[0]
        active, frozen = self.active_task, self.body_freezed
        if str(task) != str(active):
            self.assemble_model(task, freeze_body=frozen)
        with torch.no_grad():
            out = self.class_object.forward(self.model, x)
        out = out[0] if isinstance(out, (tuple, list)) else out
        if str(task) != str(active):
            self.assemble_model(active, freeze_body=frozen)
        return out.detach()

    def get_model_type(self):
        return self.model.__class__.__name__
