"""Generates tests/golden/multihead_flow_reference.json: the DYNAMIC bookkeeping of the reference's ``MultiHead_Module`` --
``update_after_iteration`` (MHM.py:139-157), ``assemble_model`` with its early return and body freezing (:326-395), ``add_new_task``
(:435-458) and the getters / setters (:488-528) -- executed step by step on the oracle network with deterministic parameter values.

After every step the sums of every tensor of the running model, the body and each head, the ``requires_grad`` flags of the running
model, the active task and ``body_freezed`` are recorded.  tests/test_host_logic.py::test_multihead_flow_matches_the_reference runs the
same script through lifelong-nnunet_amd/multihead.py.  All operations are copies and in-place additions of constants, so the sums are
exact in fp32.

    python -m oracle.make_goldens_mh_flow        (in the build container; /root/reference is not on the GPU box)

Only DATA is written (json): no reference source or bytecode is copied."""
from __future__ import annotations

import contextlib
import io
import json
import os

import torch
from torch import nn

from .make_goldens import OUT

CTOR = [1, 8, 3, 2]
SPLITS = ["seg_outputs", "tu"]


def set_deterministic(model):
    with torch.no_grad():
        for i, (n, p) in enumerate(model.named_parameters()):
            p.copy_(((i + 1) * 0.25 + (torch.arange(p.numel(), dtype=torch.float32) % 7) * 0.125).reshape(p.shape))


def snapshot(mh):
    f = lambda mod: {n: float(p.detach().double().sum()) for n, p in mod.named_parameters()}
    return {"model": f(mh.model), "body": f(mh.body), "heads": {t: f(h) for t, h in mh.heads.items()},
            "requires_grad": {n: bool(p.requires_grad) for n, p in mh.model.named_parameters()},
            "active_task": str(mh.active_task), "body_freezed": bool(mh.body_freezed)}


def add_all(mod, c):
    with torch.no_grad():
        for p in mod.parameters():
            p.add_(c)


def script(mh):
    """The scripted flow; yields (step name, snapshot).  Shared by the generator (reference class) and the test (product class)."""
    set_deterministic(mh.model)
    mh.update_after_iteration()
    yield "s0_deterministic_and_update", snapshot(mh)
    mh.add_new_task("B", use_init=False)
    yield "s1_add_task_B", snapshot(mh)
    add_all(mh.model, 1.0)
    mh.update_after_iteration()
    yield "s2_train_A_update", snapshot(mh)
    mh.assemble_model("B")
    yield "s3_assemble_B", snapshot(mh)
    add_all(mh.model, 0.5)
    mh.update_after_iteration(update_body=False)
    yield "s4_train_B_update_head_only", snapshot(mh)
    mh.assemble_model("A", freeze_body=True)
    yield "s5_assemble_A_frozen", snapshot(mh)
    b = mh.get_body()
    add_all(b, 3.0)
    mh.set_body(b)
    yield "s6_set_body", snapshot(mh)
    mh.assemble_model("A", freeze_body=True)          # same task, same flag: the reference returns early
    yield "s7_assemble_same_task_early_return", snapshot(mh)
    mh.assemble_model("B", freeze_body=True)
    yield "s8_assemble_B_takes_the_new_body", snapshot(mh)
    h = mh.get_heads()
    add_all(h["B"], 7.0)
    mh.set_heads(h, reset=False)
    mh.assemble_model("B", freeze_body=False)
    yield "s9_set_heads_assemble_unfrozen", snapshot(mh)
    b = mh.get_body()
    add_all(b, 11.0)
    mh.set_body(b)
    mh.update_after_iteration(update_body=True)
    yield "s10_set_body_dropped_by_update", snapshot(mh)
    h2 = nn.ModuleDict({"A": mh.get_heads()["A"]})
    mh.set_heads(h2, reset=True)
    yield "s11_set_heads_reset", snapshot(mh)


def main():
    from . import ref_shim
    from .unet import OracleGenericUNet
    ref_shim.install()
    from nnunet_ext.network_architecture.MultiHead_Module import MultiHead_Module
    fn = MultiHead_Module._split_model_recursively_into_body_head
    out = {}
    for sp in SPLITS:
        d = list(fn.__defaults__)        # fresh default arguments: what a fresh process sees (see make_goldens_splits.py)
        fn.__defaults__ = (nn.Module(), nn.Module(), list()) + tuple(d[3:])
        torch.manual_seed(5)
        with contextlib.redirect_stdout(io.StringIO()):
            mh = MultiHead_Module(OracleGenericUNet, sp, "A", None, *CTOR)
            out[sp] = dict(script(mh))
    json.dump({"ctor": CTOR, "flows": out}, open(os.path.join(OUT, "multihead_flow_reference.json"), "w"))
    for sp, steps in out.items():
        print(sp, list(steps))


if __name__ == "__main__":
    main()
