"""Generates tests/golden/multihead_nested_flow_reference.json: what the reference's ``MultiHead_Module`` does with a NESTED
``split_at`` (``tu.1``, ``conv_blocks_context.2``, ...) AFTER construction -- executed, step by step.

The reference re-splits the running model on every ``update_after_iteration`` (MHM.py:139-157) and its recursive split keeps its
working objects in mutable default arguments (MHM.py:159-160).  For a top-level split that is harmless; for a nested split the second
and every later call of a process starts with the path list the constructor's call left behind, never descends into the split
container, and returns (i) a body that holds EVERY top-level module and (ii) a copy of the default-argument head, which still holds
the module objects the constructor took out of the split container -- objects that left the running model at the first
``assemble_model`` and therefore keep their construction-time values.  What that means tensor by tensor is recorded here by running
two scripts on the reference class (imported from /root/reference) around the oracle network:

  * ``make_goldens_mh_flow.script`` -- the 12 steps the top-level fixture uses;
  * ``script_nested`` below -- a task added BEFORE the first re-split (a construction-time head), the first re-split with
    ``update_body=False``, that old-form head assembled into the re-split body, ``add_new_task(use_init=True)`` after a re-split
    (the reference raises: ``state_init`` has the construction-time keys), transfer of the last head, frozen body, ``set_body``.

The network is handed over as ``prev_trainer`` with deterministic values, so the construction-time values the heads keep are known
to the test without shipping tensors.  After every step: sums of every tensor of the running model / body / each head, the
``requires_grad`` flags, active task, ``body_freezed`` and a digest of the ``state_dict()`` keys (tensors are referred to by their index
in ``model.named_parameters()`` to keep the file small).
tests/test_host_logic.py::test_multihead_nested_flow_matches_the_reference runs both scripts through lifelong-nnunet_amd/multihead.py.

    python -m oracle.make_goldens_mh_nested        (in the build container; /root/reference is not on the GPU box)

Only DATA is written (json): no reference source or bytecode is copied."""
from __future__ import annotations

import contextlib
import hashlib
import io
import json
import os

import torch
from torch import nn

from .make_goldens import OUT
from .make_goldens_mh_flow import add_all, set_deterministic, snapshot
from .make_goldens_mh_flow import script as script_toplevel

CTOR = [1, 8, 3, 2]
SPLITS = ["tu.1", "seg_outputs.1", "conv_blocks_context.2", "conv_blocks_context.2.1", "conv_blocks_localization.0.1",
          "conv_blocks_context.1.blocks.1", "conv_blocks_localization.1.1.blocks.0.instnorm", "tu"]


def set_construction_values(model):
    """Deterministic values the network holds when the class is constructed around it (``prev_trainer``)."""
    with torch.no_grad():
        for i, (n, p) in enumerate(model.named_parameters()):
            p.copy_((-(i + 1) * 0.5 + (torch.arange(p.numel(), dtype=torch.float32) % 5) * 0.0625).reshape(p.shape))


def snap(mh):
    """``make_goldens_mh_flow.snapshot`` in a compact encoding: tensors by their index in ``model.named_parameters()``."""
    s = snapshot(mh)
    names = list(s["model"])
    keys = list(mh.state_dict().keys())
    return {"model": [s["model"][n] for n in names],
            "body_idx": [names.index(n) for n in s["body"]], "body": list(s["body"].values()),
            "heads": s["heads"], "frozen_idx": [i for i, n in enumerate(names) if not s["requires_grad"][n]],
            "active_task": s["active_task"], "body_freezed": s["body_freezed"],
            "state_dict_len": len(keys), "state_dict_sha256": hashlib.sha256("\n".join(keys).encode()).hexdigest()}


def with_compact_snapshots(scr):
    """Run a script of make_goldens_mh_flow (which yields its own snapshots) and re-take every snapshot in the compact encoding."""
    def run(mh):
        for name, _ in scr(mh):
            yield name, snap(mh)
    return run


def script_nested(mh):
    set_deterministic(mh.model)
    mh.add_new_task("B", use_init=True)             # before the first re-split: a head of the construction-time form
    yield "n0_add_B_before_any_update", snap(mh)
    add_all(mh.model, 1.0)
    mh.update_after_iteration(update_body=False)    # the FIRST re-split, head only
    yield "n1_first_update_head_only", snap(mh)
    mh.assemble_model("B")                          # the construction-time head into the re-split body
    yield "n2_assemble_B_old_form", snap(mh)
    add_all(mh.model, 0.5)
    mh.update_after_iteration()
    yield "n3_update_B", snap(mh)
    try:
        mh.add_new_task("C", use_init=True)
        yield "n4_add_C_use_init", snap(mh)
    except RuntimeError:
        if "C" in mh.heads:
            del mh.heads["C"]                        # the reference registers the head before load_state_dict raises
        yield "n4_add_C_use_init", {"raises": "RuntimeError"}
    mh.add_new_task("C", use_init=False)
    mh.assemble_model("C", freeze_body=True)
    yield "n5_add_C_transfer_assemble_frozen", snap(mh)
    add_all(mh.model, 0.25)
    mh.update_after_iteration()
    yield "n6_update_C_frozen", snap(mh)
    b = mh.get_body()
    add_all(b, 3.0)
    mh.set_body(b)
    mh.assemble_model("A", freeze_body=False)
    yield "n7_set_body_assemble_A", snap(mh)
    add_all(mh.model, 2.0)
    mh.update_after_iteration()
    mh.assemble_model("B")
    mh.assemble_model("A")
    yield "n8_train_A_roundtrip", snap(mh)


def main():
    from . import ref_shim
    from .unet import OracleGenericUNet
    ref_shim.install()
    from nnunet_ext.network_architecture.MultiHead_Module import MultiHead_Module
    fn = MultiHead_Module._split_model_recursively_into_body_head
    out = {}
    for sp in SPLITS:
        out[sp] = {}
        for key, scr in (("toplevel_script", with_compact_snapshots(script_toplevel)), ("nested_script", script_nested)):
            d = list(fn.__defaults__)        # fresh default arguments: what a fresh process sees (see make_goldens_splits.py)
            fn.__defaults__ = (nn.Module(), nn.Module(), list()) + tuple(d[3:])
            torch.manual_seed(5)
            net = OracleGenericUNet(*CTOR)
            set_construction_values(net)
            with contextlib.redirect_stdout(io.StringIO()):
                mh = MultiHead_Module(OracleGenericUNet, sp, "A", net, *CTOR)
                out[sp][key] = dict(scr(mh))
    names = [n for n, _ in OracleGenericUNet(*CTOR).named_parameters()]
    json.dump({"ctor": CTOR, "names": names, "flows": out}, open(os.path.join(OUT, "multihead_nested_flow_reference.json"), "w"))
    for sp, f in out.items():
        last = f["nested_script"]["n8_train_A_roundtrip"]
        print(sp, "heads", {t: len(h) for t, h in last["heads"].items()}, "body", len(last["body"]), "state_dict", last["state_dict_len"], "n4:", f["nested_script"]["n4_add_C_use_init"].get("raises", "ok"))


if __name__ == "__main__":
    main()
