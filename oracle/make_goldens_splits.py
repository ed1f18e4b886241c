"""Generates tests/golden/multihead_splits_reference.json: what the REFERENCE's ``MultiHead_Module`` does with a range of
``split_at`` strings (``MHM.py:63-137`` normalisation + ``_split_model_recursively_into_body_head``).

For every split string the reference class (imported from /root/reference, executed) is constructed around the oracle network,
a second task is added, and the following is recorded: the normalised split path, the keys of ``state_dict()`` in order, the body's and
the head's parameter names -- or, where the constructor refuses the split, that it raised an AssertionError.  (The reference's
recursive split keeps its working objects in mutable default arguments, so only the FIRST construction of a process is clean;
every entry here is generated from that clean state, see main().)
tests/test_host_logic.py::test_multihead_splits_match_the_reference holds lifelong-nnunet_amd/multihead.py to every entry.

    python -m oracle.make_goldens_splits        (in the build container; /root/reference is not on the GPU box)

Only DATA is written (json): no reference source or bytecode is copied."""
from __future__ import annotations

import contextlib
import io
import json
import os

import torch

from . import ref_shim
from .make_goldens import OUT
from .unet import OracleGenericUNet

SPLITS = ["seg_outputs", "seg_outputs.0", "seg_outputs.1", "tu", "tu.0", "tu.1", " tu . 1 ", "td",
          "conv_blocks_context", "conv_blocks_context.0", "conv_blocks_context.1", "conv_blocks_context.2",
          "conv_blocks_context.1.blocks.1", "conv_blocks_context.2.1", "conv_blocks_context.2.0",
          "conv_blocks_localization", "conv_blocks_localization.0", "conv_blocks_localization.0.0.blocks.0.conv",
          "conv_blocks_localization.1", "conv_blocks_localization.0.1", "conv_blocks_localization.1.1.blocks.0.instnorm",
          "does_not_exist", "tu.5", ""]
CTOR = [1, 8, 3, 2]


def main():
    ref_shim.install()
    from nnunet_ext.network_architecture.MultiHead_Module import MultiHead_Module
    out = {}
    from torch import nn
    fn = MultiHead_Module._split_model_recursively_into_body_head
    n_def = len(fn.__defaults__)
    for sp in SPLITS:
        torch.manual_seed(3)
        # The reference declares ``body=nn.Module(), head=nn.Module(), parent=list()`` as DEFAULT ARGUMENTS of its recursive split
        # (MHM.py:159-160): the objects are created once per process and every later construction starts from what the previous one
        # left in them (a second ``MultiHead_Module`` in the same process raises or returns a wrong body/head).  Each entry below is
        # what a FRESH process sees: the defaults are re-created before every construction.
        d = list(fn.__defaults__)
        assert n_def == 4 and isinstance(d[0], nn.Module) and isinstance(d[1], nn.Module) and isinstance(d[2], list), d
        fn.__defaults__ = (nn.Module(), nn.Module(), list()) + tuple(d[3:])
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                mh = MultiHead_Module(OracleGenericUNet, sp, "taskA", None, *CTOR)
                mh.add_new_task("taskB", use_init=True)
        except AssertionError:
            out[sp] = {"raises": "AssertionError"}
            continue
        except Exception as e:      # anything else the reference does with this string is recorded by type
            out[sp] = {"raises": type(e).__name__}
            continue
        out[sp] = {"split": ".".join(mh.split), "state_dict_keys": list(mh.state_dict().keys()),
                   "body_param_names": [n for n, _ in mh.body.named_parameters()],
                   "head_param_names": [n for n, _ in mh.heads["taskA"].named_parameters()],
                   "model_param_names": [n for n, _ in mh.model.named_parameters()]}
    json.dump({"ctor": CTOR, "splits": out}, open(os.path.join(OUT, "multihead_splits_reference.json"), "w"), indent=1)
    for sp, r in out.items():
        print(repr(sp), "->", r.get("raises") or (r["split"], len(r["body_param_names"]), len(r["head_param_names"]), r["head_param_names"][:2]))


if __name__ == "__main__":
    main()
