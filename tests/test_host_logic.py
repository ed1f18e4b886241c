"""CPU tests of the host-side mirrors (no kernel launches): C-ABI symbol coverage, MultiHead_Module semantics
and state-dict naming against the reference-generated golden, trainer plugin surface, rehearsal sampling."""
import json
import os
import re

import numpy as np
import pytest
import torch

import lifelong_nnunet_amd as pkg
from lifelong_nnunet_amd import native as nat
from lifelong_nnunet_amd.multihead import MultiHead_Module
from lifelong_nnunet_amd.network import Generic_UNet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _headers():
    """The public C-ABI (include/lnn_hip.h) followed by the debug / test hooks (csrc/lnn_debug.h, not part of the drop-in surface)."""
    return (open(os.path.join(ROOT, "include", "lnn_hip.h")).read(),
            open(os.path.join(ROOT, "lifelong-nnunet_amd", "csrc", "lnn_debug.h")).read())


def test_cabi_exports_every_declared_symbol():
    pub, dbg = _headers()
    assert "lnn_debug_" not in pub, "debug hooks belong in csrc/lnn_debug.h, not in the public header"
    hdr = pub + dbg
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(lnn_[a-zA-Z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = nat.lib()                                   # loads, no GPU needed
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/lnn_hip.h but not exported"
        assert name in nat.SIGNATURES, f"{name} has no ctypes signature"
    assert set(nat.SIGNATURES) == declared
    assert lib.lnn_version() >= 100


def test_ctypes_signatures_match_the_header():
    """Every declaration of include/lnn_hip.h against the ctypes table: same number of parameters, and per parameter the same
    class (pointer / int / long / float / size_t) -- a wrong table entry would corrupt a call silently."""
    import ctypes as C
    hdr = "".join(_headers())
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    hdr = re.sub(r"//[^\n]*", "", hdr)
    decls = re.findall(r"\b([a-zA-Z_][a-zA-Z0-9_ \*]*?)\s*\b(lnn_[a-zA-Z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S)
    assert len(decls) >= 30

    def klass(param):
        param = " ".join(param.split())
        if "*" in param or param.startswith("lnn_stream_t"):
            return C.c_void_p
        base = param.rsplit(" ", 1)[0] if " " in param else param
        return {"int": C.c_int, "long": C.c_long, "float": C.c_float, "size_t": C.c_size_t, "unsigned": C.c_uint}[base.replace("const ", "")]

    same = {C.c_char_p: C.c_void_p}
    for ret, name, params in decls:
        res, argtypes = nat.SIGNATURES[name]
        plist = [q for q in (x.strip() for x in params.split(",")) if q and q != "void"]
        assert len(plist) == len(argtypes), f"{name}: header has {len(plist)} parameters, ctypes table {len(argtypes)}"
        for i, (q, a) in enumerate(zip(plist, argtypes)):
            want = klass(q)
            got = same.get(a, a)
            if hasattr(got, "_type_") and not isinstance(got._type_, str):       # POINTER(c_int) and friends
                got = C.c_void_p
            assert got is want, f"{name}: parameter {i} ({q!r}) is {want.__name__} in the header, {a} in the ctypes table"


def test_bad_arguments_are_reported_not_thrown():
    lib = nat.lib()
    rc = lib.lnn_pack_weights(None, None, None, 27, 32, 32, 1, 1, 1)
    assert rc == -1 and b"lnn_pack_weights" in lib.lnn_last_error()
    rc = lib.lnn_conv3d_fwd(None, None, 32, None, None, None, 32, 1, 8, 8, 8, 32, 32, 3)
    assert rc == -1 and b"stride" in lib.lnn_last_error()


def test_generic_unet_parameter_names_match_reference(golden_dir):
    m = json.load(open(f"{golden_dir}/meta.json"))
    net = Generic_UNet(*m["toy_unet"]["ctor"], device="cpu")
    assert [n for n, _ in net.named_parameters()] == m["toy_unet"]["param_names"]
    assert [n for n, _ in net.named_parameters()] == m["multihead"]["model_param_names"]
    # every parameter / gradient is a view into the flat arenas
    for n, p in net.named_parameters():
        assert p.data_ptr() == net.arena.theta.data_ptr() + 4 * p._lnn_slot.offset
        assert p.grad.data_ptr() == net.arena.grad.data_ptr() + 4 * p._lnn_slot.offset
    # He init: std = sqrt(2/(1+a^2)) / sqrt(fan_in); ConvTranspose fan_in uses dim 1
    big = Generic_UNet(1, 32, 3, 3, device="cpu")
    sd = big.state_dict()
    w = sd["conv_blocks_context.1.blocks.1.conv.weight"]
    assert abs(float(w.std()) - 1.41414 / (64 * 27) ** 0.5) < 0.02 * float(w.std())
    tu = sd["tu.0.weight"]
    assert abs(float(tu.std()) - 1.41414 / (tu.shape[1] * 8) ** 0.5) < 0.03 * float(tu.std())
    assert float(sd["conv_blocks_context.0.blocks.0.instnorm.weight"].min()) == 1.0
    assert float(sd["conv_blocks_context.0.blocks.0.conv.bias"].abs().max()) == 0.0


def test_multihead_module_matches_reference_naming_and_semantics(golden_dir):
    m = json.load(open(f"{golden_dir}/meta.json"))["multihead"]
    mh = MultiHead_Module(Generic_UNet, "seg_outputs", "taskA", None, *m["ctor"], device="cpu")
    mh.add_new_task("taskB", use_init=False)
    assert list(mh.state_dict().keys()) == m["state_dict_keys"]
    assert [n for n, _ in mh.body.named_parameters()] == m["body_param_names"]
    assert [n for n, _ in mh.heads["taskA"].named_parameters()] == m["head_param_names"]
    # body tensors are SHARED with the running model, head tensors are copies
    body = dict(mh.body.named_parameters()); model = dict(mh.model.named_parameters())
    assert all(body[n] is model[n] for n in body)
    hA = dict(mh.heads["taskA"].named_parameters())
    assert all(hA[n] is not model[n] and torch.equal(hA[n], model[n]) for n in hA)
    # training changes the running model; update_after_iteration refreshes ONLY the active head
    with torch.no_grad():
        for n in hA:
            model[n].add_(1.0)
    mh.update_after_iteration()
    hB = dict(mh.heads["taskB"].named_parameters())
    assert all(torch.equal(hA[n], model[n]) for n in hA)
    assert all(not torch.equal(hB[n], model[n]) for n in hB)
    mh.assemble_model("taskB")
    assert mh.active_task == m["active_after_assemble"] == "taskB"
    assert all(torch.equal(hB[n], model[n]) for n in hB)
    mh.assemble_model("taskB", freeze_body=True)
    assert all(not model[n].requires_grad for n in body) and all(model[n].requires_grad for n in hB)
    mh.assemble_model("taskB", freeze_body=False)
    assert all(model[n].requires_grad for n in body)
    # split normalisation (MHM.py:73-92) and invalid splits (test_MultiHead_Module.py:198-270)
    mh2 = MultiHead_Module(Generic_UNet, "seg_outputs.0", "t", None, *m["ctor"], device="cpu")
    assert mh2.get_split_path() == "seg_outputs"
    with pytest.raises(AssertionError):
        MultiHead_Module(Generic_UNet, "does_not_exist", "t", None, *m["ctor"], device="cpu")
    with pytest.raises(AssertionError):
        MultiHead_Module(Generic_UNet, "conv_blocks_localization", "t", None, *m["ctor"], device="cpu")
    with pytest.raises(AssertionError):
        mh.assemble_model("unknown_task")
    # deeper split: everything at or after the split point is head
    mh3 = MultiHead_Module(Generic_UNet, "tu", "t", None, *m["ctor"], device="cpu")
    assert [n for n, _ in mh3.heads["t"].named_parameters()] == ["tu.0.weight", "tu.1.weight", "seg_outputs.0.weight", "seg_outputs.1.weight"]


def test_multihead_splits_match_the_reference(golden_dir):
    """tests/golden/multihead_splits_reference.json: the reference's ``MultiHead_Module`` CONSTRUCTED (oracle/make_goldens_splits.py)
    with 24 ``split_at`` strings -- top-level and nested paths, paths that simplify (``tu.0`` -> ``tu``), a parameter-less module
    (``td``), whitespace, paths before the first layer and paths that do not exist.  The product must agree on: refusal
    (AssertionError), the normalised split, the ``state_dict()`` key list in order, body and head parameter names.  That includes the
    reference's behaviour for NESTED splits: the body keeps the whole top-level container of the split, so its head-side members
    are body AND head."""
    g = json.load(open(f"{golden_dir}/multihead_splits_reference.json"))
    assert len(g["splits"]) == 24
    for sp, r in g["splits"].items():
        if "raises" in r:
            assert r["raises"] == "AssertionError", (sp, r)
            with pytest.raises(AssertionError):
                MultiHead_Module(Generic_UNet, sp, "taskA", None, *g["ctor"], device="cpu")
            continue
        mh = MultiHead_Module(Generic_UNet, sp, "taskA", None, *g["ctor"], device="cpu")
        mh.add_new_task("taskB", use_init=True)
        assert mh.get_split_path() == r["split"], sp
        assert list(mh.state_dict().keys()) == r["state_dict_keys"], sp
        assert [n for n, _ in mh.body.named_parameters()] == r["body_param_names"], sp
        assert [n for n, _ in mh.heads["taskA"].named_parameters()] == r["head_param_names"], sp
        assert [n for n, _ in mh.model.named_parameters()] == r["model_param_names"], sp
        # body tensors are the running model's own, head tensors are copies -- also where a nested split makes a tensor both
        body, model = dict(mh.body.named_parameters()), dict(mh.model.named_parameters())
        assert all(body[n] is model[n] for n in body), sp
        assert all(p is not model[n] for n, p in mh.heads["taskA"].named_parameters()), sp
        mh.assemble_model("taskB", freeze_body=True)         # MHM.py:379-395: every body NAME is frozen in the running model
        assert all(model[n].requires_grad == (n not in body) for n in model), sp
    nested = g["splits"]["tu.1"]
    assert "tu.1.weight" in nested["body_param_names"] and "tu.1.weight" in nested["head_param_names"]


def test_multihead_flow_matches_the_reference(golden_dir):
    """tests/golden/multihead_flow_reference.json: a 12-step script -- deterministic parameters, update_after_iteration (with and without
    the body), add_new_task, assemble_model incl. its early return and body freezing, get_body / set_body (the running model only sees a
    new body at the next assemble_model that does not return early; update_after_iteration drops it), get_heads / set_heads (update and
    reset) -- EXECUTED on the reference's ``MultiHead_Module`` (oracle/make_goldens_mh_flow.py) for the two top-level splits
    ``seg_outputs`` and ``tu``; the same script through the product must leave the same tensors in the running model, the body and every
    head, the same ``requires_grad`` flags, active task and ``body_freezed`` after every step."""
    from oracle.make_goldens_mh_flow import script
    g = json.load(open(f"{golden_dir}/multihead_flow_reference.json"))
    close = lambda a, b: abs(a - b) <= 1e-6 * max(1.0, abs(b))
    for sp, steps in g["flows"].items():
        mh = MultiHead_Module(Generic_UNet, sp, "A", None, *g["ctor"], device="cpu")
        seen = []
        for name, snap in script(mh):
            ref = steps[name]
            seen.append(name)
            for part in ("model", "body"):
                assert list(snap[part]) == list(ref[part]), (sp, name, part)
                assert all(close(snap[part][n], ref[part][n]) for n in ref[part]), (sp, name, part)
            assert list(snap["heads"]) == list(ref["heads"]), (sp, name)
            for t in ref["heads"]:
                assert list(snap["heads"][t]) == list(ref["heads"][t]), (sp, name, t)
                assert all(close(snap["heads"][t][n], ref["heads"][t][n]) for n in ref["heads"][t]), (sp, name, t)
            assert snap["requires_grad"] == ref["requires_grad"], (sp, name)
            assert snap["active_task"] == ref["active_task"] and snap["body_freezed"] == ref["body_freezed"], (sp, name)
        assert seen == list(steps) and len(seen) == 12


def test_multihead_nested_flow_matches_the_reference(golden_dir):
    """tests/golden/multihead_nested_flow_reference.json: NESTED splits after construction.  Two scripts (the 12 steps of the top-level
    fixture; 9 steps with a task added before the first re-split, a head-only first update, ``add_new_task(use_init=True)`` after a
    re-split -- the reference raises --, head transfer, frozen body, ``set_body``) EXECUTED on the reference's ``MultiHead_Module``
    (oracle/make_goldens_mh_nested.py) for seven nested splits of depth 2-6 and one top-level split.  The product must leave the same
    tensors in the running model, the body and every head, the same frozen set, flags and ``state_dict()`` keys after every step: from
    the first ``update_after_iteration`` on every tensor is body and the active head is the innermost split container's tail at its
    construction-time values (multihead.py:_reference_resplit)."""
    import hashlib
    from oracle.make_goldens_mh_flow import script as script_toplevel
    from oracle.make_goldens_mh_nested import script_nested, set_construction_values
    g = json.load(open(f"{golden_dir}/multihead_nested_flow_reference.json"))
    names = g["names"]
    close = lambda a, b: abs(a - b) <= 1e-6 * max(1.0, abs(b))
    nsteps = 0
    for sp, flows in g["flows"].items():
        for key, scr in (("toplevel_script", script_toplevel), ("nested_script", script_nested)):
            net = Generic_UNet(*g["ctor"], device="cpu")
            assert [n for n, _ in net.named_parameters()] == names
            set_construction_values(net)
            mh = MultiHead_Module(Generic_UNet, sp, "A", net)
            seen = []
            for name, _ in scr(mh):
                ref, where = flows[key][name], (sp, key, name)
                seen.append(name)
                if "raises" in ref:
                    assert ref["raises"] == "RuntimeError" and "C" not in mh.heads, where      # the script removed the refused head
                    continue
                model = dict(mh.model.named_parameters())
                val = lambda p: float(p.detach().double().sum())
                assert list(model) == names and all(close(val(model[n]), r) for n, r in zip(names, ref["model"])), where
                body = dict(mh.body.named_parameters())
                assert list(body) == [names[i] for i in ref["body_idx"]], where
                assert all(close(val(p), r) for p, r in zip(body.values(), ref["body"])), where
                assert list(mh.heads.keys()) == list(ref["heads"]), where
                for t, h in ref["heads"].items():
                    mine = dict(mh.heads[t].named_parameters())
                    assert list(mine) == list(h) and all(close(val(mine[n]), h[n]) for n in h), where + (t,)
                assert [i for i, n in enumerate(names) if not model[n].requires_grad] == ref["frozen_idx"], where
                assert str(mh.active_task) == ref["active_task"] and bool(mh.body_freezed) == ref["body_freezed"], where
                keys = list(mh.state_dict().keys())
                assert len(keys) == ref["state_dict_len"] and hashlib.sha256("\n".join(keys).encode()).hexdigest() == ref["state_dict_sha256"], where
                nsteps += 1
            assert seen == list(flows[key])
    assert nsteps >= 8 * 20
    # the switch: with reference_nested_resplit off the construction-time partition survives the updates
    try:
        MultiHead_Module.reference_nested_resplit = False
        mh = MultiHead_Module(Generic_UNet, "tu.1", "A", None, *g["ctor"], device="cpu")
        h0 = [n for n, _ in mh.heads["A"].named_parameters()]
        with torch.no_grad():
            dict(mh.model.named_parameters())["tu.1.weight"].add_(1.0)
        mh.update_after_iteration()
        assert [n for n, _ in mh.heads["A"].named_parameters()] == h0 == ["tu.1.weight", "seg_outputs.0.weight", "seg_outputs.1.weight"]
        assert torch.equal(dict(mh.heads["A"].named_parameters())["tu.1.weight"], dict(mh.model.named_parameters())["tu.1.weight"])
    finally:
        MultiHead_Module.reference_nested_resplit = True


def test_network_configuration_matches_the_reference(golden_dir):
    """tests/golden/network_config_reference.json: the constructor arguments the reference's ``initialize_network``
    (nnViTUNetTrainer.py:97-122) EXECUTED with a recording network class produces.  The oracle network and the product's constants must
    be that configuration: Conv3d / InstanceNorm3d(eps 1e-5, affine) / no dropout / LeakyReLU(1e-2), deep supervision, identity final
    nonlinearity, He initialisation with a = 1e-2, convolutional pooling and upsampling, no logits upscaling, feature doubling capped at
    320, two convs per stage."""
    from torch import nn
    from oracle.unet import OracleGenericUNet
    from lifelong_nnunet_amd import engine
    c = json.load(open(f"{golden_dir}/network_config_reference.json"))
    assert c["conv_op"].endswith("Conv3d") and c["norm_op"].endswith("InstanceNorm3d") and c["nonlin"].endswith("LeakyReLU")
    assert c["norm_op_kwargs"] == {"eps": 1e-5, "affine": True} and c["dropout_op_kwargs"]["p"] == 0
    assert c["nonlin_kwargs"]["negative_slope"] == 1e-2 and c["he_init_neg_slope"] == 1e-2
    assert c["deep_supervision"] is True and c["dropout_in_localization"] is False and c["upscale_logits"] is False
    assert c["convolutional_pooling"] is True and c["convolutional_upsampling"] is True
    assert c["final_nonlin_of_probe"] == [-2.0, 0.5] and c["num_conv_per_stage"] == 2 and c["feat_map_mul_on_downscale"] == 2
    # product constants
    assert engine.LRELU_SLOPE == c["nonlin_kwargs"]["negative_slope"] and engine.IN_EPS == c["norm_op_kwargs"]["eps"]
    # oracle network built with the same arguments
    net = OracleGenericUNet(c["input_channels"], 4, c["num_classes"], 3, conv_per_stage=c["num_conv_per_stage"])
    norms = [m for m in net.modules() if isinstance(m, nn.InstanceNorm3d)]
    acts = [m for m in net.modules() if isinstance(m, nn.LeakyReLU)]
    assert norms and all(m.eps == 1e-5 and m.affine for m in norms) and acts and all(m.negative_slope == 1e-2 for m in acts)
    assert not any(isinstance(m, (nn.Dropout3d, nn.Dropout, nn.MaxPool3d, nn.Upsample)) for m in net.modules())
    convs = [m for m in net.modules() if isinstance(m, nn.Conv3d) and m.kernel_size == (3, 3, 3)]
    assert all(m.padding == (1, 1, 1) for m in convs) and {m.stride for m in convs} == {(1, 1, 1), (2, 2, 2)}      # convolutional pooling
    assert len(net.td) == 0 and all(isinstance(m, nn.ConvTranspose3d) and m.kernel_size == (2, 2, 2) and m.stride == (2, 2, 2)
                                    and m.bias is None for m in net.tu)                                           # convolutional upsampling
    assert net._deep_supervision and net.do_ds and len(net.seg_outputs) == 3
    assert all(m.kernel_size == (1, 1, 1) and m.bias is None for m in net.seg_outputs)
    # the product network has the same parameters (names, shapes) as the oracle network built from that configuration
    prod = Generic_UNet(c["input_channels"], 4, c["num_classes"], 3, device="cpu")
    assert [(n, tuple(p.shape)) for n, p in prod.named_parameters()] == [(n, tuple(p.shape)) for n, p in net.named_parameters()]
    assert [ch for ch in (c["base_num_features"] * 2 ** d for d in range(6))][:3] == [32, 64, 128] and OracleGenericUNet.MAX_FEATURES_3D == 320


def test_anisotropic_plan_builds_the_reference_configuration(golden_dir):
    """A plan with two modalities and per-axis poolings / kernels (Task005_Prostate-shaped, README.md:73): the per-level lists the
    REFERENCE's ``initialize_network`` hands to its network class (``prostate_shaped`` of network_config_reference.json, recorded by
    executing nnViTUNetTrainer.py:97-122) build, in the product and in the oracle, networks with the same parameter names and shapes
    in the same order; shapes follow upstream's indexing (decoder stage u: transposed conv = pooling -(u+1), conv kernel -(u+1))."""
    from oracle.unet import OracleGenericUNet
    from lifelong_nnunet_amd.engine import unet_geometry
    from lifelong_nnunet_amd.synthetic import ds_strides, make_patch_batch
    c = json.load(open(f"{golden_dir}/network_config_reference.json"))["prostate_shaped"]
    assert c["input_channels"] == 2 and c["num_pool"] == len(c["pool_op_kernel_sizes"]) == 3 and len(c["conv_kernel_sizes"]) == 4
    args = (c["input_channels"], c["base_num_features"], c["num_classes"], c["num_pool"])
    kw = dict(pool_op_kernel_sizes=c["pool_op_kernel_sizes"], conv_kernel_sizes=c["conv_kernel_sizes"])
    onet = OracleGenericUNet(*args, conv_per_stage=c["num_conv_per_stage"], **kw)
    prod = Generic_UNet(*args, device="cpu", **kw)
    assert [(n, tuple(p.shape)) for n, p in prod.named_parameters()] == [(n, tuple(p.shape)) for n, p in onet.named_parameters()]
    sh = {n: tuple(p.shape) for n, p in prod.named_parameters()}
    assert sh["conv_blocks_context.0.blocks.0.conv.weight"] == (8, 2, 1, 3, 3)
    assert sh["conv_blocks_context.2.blocks.0.conv.weight"] == (32, 16, 3, 3, 3)
    assert sh["tu.0.weight"] == (64, 32, 2, 2, 2) and sh["tu.2.weight"] == (16, 8, 1, 2, 2)
    assert sh["conv_blocks_localization.2.0.blocks.0.conv.weight"] == (8, 16, 1, 3, 3)      # kernel -(u+1) = kernel 1 for u = 2
    assert sh["conv_blocks_localization.1.0.blocks.0.conv.weight"] == (16, 32, 3, 3, 3)     # ... = kernel 2 for u = 1
    # the oracle's forward runs on the plan's patch and yields the deep-supervision resolutions of the cumulative poolings
    x, tg = make_patch_batch(1, c["patch_size"], c["num_pool"], in_channels=2, pool_op_kernel_sizes=c["pool_op_kernel_sizes"])
    with torch.no_grad():
        outs = onet(x)
    pools, kernels, dims = unet_geometry(c["num_pool"], c["patch_size"], **kw)
    assert [tuple(o.shape[2:]) for o in outs] == dims[:3] == [(8, 32, 32), (8, 16, 16), (8, 8, 8)]
    assert [tuple(t.shape[2:]) for t in tg] == dims[:3] and ds_strides(3, c["pool_op_kernel_sizes"]) == [(1, 1, 1), (1, 2, 2), (1, 4, 4)]
    # defaults are the isotropic plan
    assert unet_geometry(2, (8, 8, 8)) == ([(2, 2, 2)] * 2, [(3, 3, 3)] * 3, [(8, 8, 8), (4, 4, 4), (2, 2, 2)])
    with pytest.raises(AssertionError):
        unet_geometry(2, (8, 6, 8))          # not divisible by the cumulative pooling


def test_trainer_plugin_surface():
    for ext, cls_name, hp in (("sequential", "nnUNetTrainerSequential", {}), ("ewc", "nnUNetTrainerEWC", {"ewc_lambda": float}),
                              ("lwf", "nnUNetTrainerLWF", {"lwf_temperature": float}),
                              ("rehearsal", "nnUNetTrainerRehearsal", {"samples_in_perc": float, "seed": int}),
                              ("rehearsal_ewc", "nnUNetTrainerRehearsalEWC", {"ewc_lambda": float, "samples_in_perc": float, "seed": int})):
        cls = pkg.get_trainer_class(ext)
        assert cls.__name__ == cls_name
        mod = __import__(cls.__module__, fromlist=["HYPERPARAMS"])
        assert mod.HYPERPARAMS == hp
        for meth in ("initialize", "run_training", "run_iteration", "reinitialize", "_perform_validation",
                     "save_checkpoint", "load_checkpoint_ram"):
            assert callable(getattr(cls, meth))
    import inspect
    assert inspect.signature(pkg.get_trainer_class("ewc").__init__).parameters["ewc_lambda"].default == 0.4
    assert inspect.signature(pkg.get_trainer_class("lwf").__init__).parameters["lwf_temperature"].default == 2.0
    sig = inspect.signature(pkg.get_trainer_class("rehearsal").__init__).parameters
    assert sig["samples_in_perc"].default == 0.25 and sig["seed"].default == 3299


def test_rehearsal_sampling_semantics(golden_dir):
    """Default (synthetic) provider: the fused training list = the current task's fold-0 training cases followed by a
    seeded 25 % sample of each previous head's training cases, drawn in head order from ONE random.seed(3299) stream
    (REH.py:73,132; the reference-generated fixture is checked in tests/test_reference_trainer_goldens.py)."""
    import random
    from collections import OrderedDict
    from lifelong_nnunet_amd.dataloading import do_split
    from lifelong_nnunet_amd.training.network_training.rehearsal.nnUNetTrainerRehearsal import nnUNetTrainerRehearsal, task_cases

    class _MH:
        heads = OrderedDict([("taskA", None), ("taskB", None)])
    tr = nnUNetTrainerRehearsal("seg_outputs", "taskC", device="cpu")
    tr.mh_network = _MH()
    tr.tr_gen, tr.val_gen = tr.get_basic_generators()
    split = lambda t: list(do_split(OrderedDict((k, 0) for k in task_cases(t, 40)), 0)[0].keys())
    random.seed(3299)
    expA = random.sample(split("taskA"), 8)
    expB = random.sample(split("taskB"), 8)
    assert tr.sampled == {"taskA": expA, "taskB": expB}
    assert list(tr.dataset_tr.keys()) == split("taskC") + expA + expB
    assert len(tr.dataset_val) == 8 and not set(tr.dataset_val) & set(tr.dataset_tr)
    batch = next(tr.tr_gen)
    assert tuple(batch["data"].shape) == (2, 1, 40, 56, 40) and len(batch["target"]) == 3 and len(batch["keys"]) == 2


def test_sliding_window_steps_and_gaussian_map():
    """Host logic of the tiled predictor (lifelong-nnunet_amd/inference.py) against hand-computed cases of the upstream
    formulas and against the independent restatement in oracle/inference.py."""
    import numpy as np
    from lifelong_nnunet_amd.inference import compute_steps_for_sliding_window, dice_per_class, get_gaussian, pad_to_patch
    from oracle import inference as oinf
    # patch 16, step 0.5 -> target step 8 voxels: image 24 -> ceil(8/8)+1 = 2 tiles at 0, 8; image 16 -> one tile
    assert compute_steps_for_sliding_window((16, 16, 16), (24, 16, 40), 0.5) == [[0, 8], [0], [0, 8, 16, 24]]
    # non-divisible span: image 29 -> ceil(13/8)+1 = 3 tiles over [0, 13] -> 0, round(6.5) = 6 (banker's), 13
    assert compute_steps_for_sliding_window((16,), (29,), 0.5) == [[0, 6, 13]]
    assert compute_steps_for_sliding_window((160, 192, 160), (160, 192, 160), 0.5) == [[0], [0], [0]]
    for ps, im, st in [((16, 16, 16), (24, 40, 24), 0.5), ((8, 12, 8), (31, 12, 9), 0.25), ((4, 4, 4), (4, 5, 11), 1.0)]:
        assert compute_steps_for_sliding_window(ps, im, st) == oinf.steps_for_sliding_window(ps, im, st)
    g = get_gaussian((16, 24, 16))
    assert g.shape == (16, 24, 16) and g.dtype == np.float32
    assert g.max() == 1.0 and np.unravel_index(g.argmax(), g.shape) == (8, 12, 8) and g.min() > 0
    assert np.array_equal(g, oinf.gaussian_map((16, 24, 16)))
    # separable: the map along an axis through the centre is exp(-d^2 / (2 sigma^2)) up to the boundary truncation
    line = g[:, 12, 8].astype(np.float64)
    sigma = 16 / 8
    assert abs(line[8 + 2] / line[8] - np.exp(-2.0 ** 2 / (2 * sigma ** 2))) < 2e-2
    x = torch.arange(2 * 3 * 5 * 4, dtype=torch.float32).reshape(2, 3, 5, 4)
    padded, slicer = pad_to_patch(x, (6, 4, 7))
    assert tuple(padded.shape) == (2, 6, 5, 7) and slicer == (slice(1, 4), slice(0, 5), slice(1, 5))
    assert torch.equal(padded[(slice(None),) + slicer], x) and float(padded.sum()) == float(x.sum())
    seg = np.array([[0, 1, 1, 2]]); lab = np.array([[0, 1, 2, 2]])
    d = dice_per_class(seg, lab, 4)
    assert d[1]["Dice"] == 2 / 3 and d[2]["Dice"] == 2 / 3 and d[1]["IoU"] == 0.5 and np.isnan(d[3]["Dice"])


def test_checkpoint_interoperates_with_torch_optimizer_and_upstream_layout(tmp_path, golden_dir):
    """SURVEY.md 8f rank 4 (checkpoint interop): ``save_checkpoint`` writes the dictionary upstream
    ``NetworkTrainer.load_checkpoint_ram`` consumes -- MultiHead_Module state-dict keys as the REFERENCE class produces
    them (golden), ``torch.optim.SGD`` / ``GradScaler`` state layouts, ``epoch + 1``, the ``.pkl`` sidecar -- and reads
    the same back, including a state dict written by a real ``torch.optim.SGD``."""
    import pickle
    from lifelong_nnunet_amd import get_trainer_class
    from oracle.unet import OracleGenericUNet
    plans = {"patch_size": (16, 16, 16), "batch_size": 2, "num_pool": 2, "base_num_features": 8, "num_classes": 3,
             "num_input_channels": 1, "synthetic_period": 4}
    tr = get_trainer_class("multihead")("seg_outputs", "taskA", plans=plans, device="cpu")
    tr.initialize(True, num_epochs=3)
    tr.mh_network.add_new_task("taskB", use_init=False)
    tr.epoch = 2
    tr.all_tr_losses, tr.all_val_losses = [0.5, 0.4], [0.6]
    g = torch.Generator().manual_seed(1)
    tr.network.arena.momentum.copy_(torch.randn(tr.network.arena.size, generator=g))
    fname = str(tmp_path / "model_final_checkpoint.model")
    ckpt = tr.save_checkpoint(fname)
    # ---- layout
    m = json.load(open(f"{golden_dir}/meta.json"))["multihead"]
    assert list(ckpt["state_dict"].keys()) == m["state_dict_keys"]          # produced by the reference MultiHead_Module
    assert ckpt["epoch"] == 3 and ckpt["lr_scheduler_state_dict"] is None and len(ckpt["plot_stuff"]) == 4
    assert set(ckpt["amp_grad_scaler"]) == {"scale", "growth_factor", "backoff_factor", "growth_interval", "_growth_tracker"}
    info = pickle.load(open(fname + ".pkl", "rb"))
    assert info["name"] == "nnUNetTrainerMultiHead" and info["plans"]["num_pool"] == 2 and info["init"][0] == "seg_outputs"
    disk = torch.load(fname, weights_only=False)
    assert list(disk["state_dict"].keys()) == list(ckpt["state_dict"].keys())
    # ---- a real torch.optim.SGD over a network with the same parameters accepts the optimiser state ...
    onet = OracleGenericUNet(1, 8, 3, 2)
    names = [n for n, p in onet.named_parameters() if p.requires_grad]
    assert names == [n for n, _ in tr.optimizer._trainable()]
    topt = torch.optim.SGD([p for p in onet.parameters() if p.requires_grad], 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)
    topt.load_state_dict(ckpt["optimizer_state_dict"])
    by_name = dict(tr.network.named_parameters())
    for p_, n in zip(topt.param_groups[0]["params"], names):
        s = by_name[n]._lnn_slot
        assert torch.equal(topt.state[p_]["momentum_buffer"].reshape(-1), tr.network.arena.momentum[s.offset:s.offset + s.numel])
    # ... and its own state dict loads back into the fused optimiser (a checkpoint written by a reference trainer)
    orig0 = ckpt["optimizer_state_dict"]["state"][0]["momentum_buffer"].clone()      # torch may alias the loaded tensors
    orig3 = ckpt["optimizer_state_dict"]["state"][3]["momentum_buffer"].clone()
    for st in topt.state.values():
        st["momentum_buffer"].mul_(2.0)
    topt.param_groups[0]["lr"] = 3e-3
    tr.optimizer.load_state_dict(topt.state_dict())
    assert tr.optimizer.param_groups[0]["lr"] == 3e-3
    s0 = by_name[names[0]]._lnn_slot
    assert torch.equal(tr.network.arena.momentum[s0.offset:s0.offset + s0.numel],
                       2.0 * orig0.reshape(-1))
    # ---- round trip through a fresh trainer (upstream strips DataParallel's "module." prefix: do the same)
    tr2 = get_trainer_class("multihead")("seg_outputs", "taskA", plans=plans, device="cpu")
    tr2.initialize(True, num_epochs=3)
    ck2 = dict(disk)
    ck2["state_dict"] = {"module." + k: v for k, v in disk["state_dict"].items()}
    tr2.load_checkpoint_ram(ck2, train=True)
    assert tr2.epoch == 3 and list(tr2.mh_network.heads.keys()) == ["taskA", "taskB"] and tr2.all_tr_losses == [0.5, 0.4]
    for (k1, v1), (k2, v2) in zip(tr.mh_network.state_dict().items(), tr2.mh_network.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    s = by_name[names[3]]._lnn_slot
    assert torch.equal(tr2.network.arena.momentum[s.offset:s.offset + s.numel], orig3.reshape(-1))
    assert tr2.amp_grad_scaler.get_scale() == tr.amp_grad_scaler.get_scale()


def _write_case(folder, name, shape, seed, channels=1):
    import pickle
    rng = np.random.RandomState(seed)
    img = rng.randn(channels, *shape).astype(np.float32)
    seg = np.zeros(shape, dtype=np.float32)
    c = [s // 2 for s in shape]
    seg[c[0] - 2:c[0] + 2, c[1] - 3:c[1] + 3, c[2] - 2:c[2] + 2] = 1
    seg[c[0] - 1:c[0] + 1, c[1] - 1:c[1] + 1, c[2] - 1:c[2] + 1] = 2
    np.savez(os.path.join(folder, name + ".npz"), data=np.concatenate([img, seg[None]], 0))
    locs = {k: np.argwhere(seg == k) for k in (1, 2)}
    pickle.dump({"class_locations": locs, "size_after_resampling": shape}, open(os.path.join(folder, name + ".pkl"), "wb"))
    return img, seg


def test_preprocessed_data_loader(tmp_path):
    """lifelong-nnunet_amd/dataloading.py (restated upstream loader -- parity unpinned): crops are
    exact sub-volumes of their case, foreground oversampling, padding values, split, deep-supervision targets."""
    from lifelong_nnunet_amd.dataloading import (DataLoader3D, PreprocessedDataProvider, do_split, downsample_seg_for_ds,
                                                 load_dataset, unpack_dataset)
    folder = str(tmp_path / "Task900_Toy" / "nnUNetData_plans_v2.1_stage0")
    os.makedirs(folder)
    cases = {f"toy_{i:03d}": _write_case(folder, f"toy_{i:03d}", (20 + 2 * i, 24, 18 + i), i) for i in range(10)}
    ds = load_dataset(folder)
    assert list(ds.keys()) == sorted(cases) and "class_locations" in ds["toy_000"]["properties"]
    unpack_dataset(folder)
    assert os.path.isfile(os.path.join(folder, "toy_003.npy"))
    tr, val = do_split(ds, 0)
    from sklearn.model_selection import KFold
    keys = np.array(sorted(cases))
    ref_tr, ref_val = next(iter(KFold(n_splits=5, shuffle=True, random_state=12345).split(keys)))
    assert list(tr) == sorted(keys[ref_tr]) and list(val) == sorted(keys[ref_val]) and not set(tr) & set(val)
    # ---- every crop is an exact sub-volume; the last round(B * 0.33) samples contain foreground
    np.random.seed(0)
    B, ps = 6, (16, 16, 16)
    dl = DataLoader3D(ds, ps, ps, B, False, oversample_foreground_percent=0.33, pad_mode="constant", memmap_mode="r")
    assert [dl.get_do_oversample(j) for j in range(B)] == [False] * 4 + [True] * 2
    for _ in range(5):
        b = next(dl)
        assert b["data"].shape == (B, 1) + ps and b["seg"].shape == (B, 1) + ps
        for j, k in enumerate(b["keys"]):
            img, seg = cases[k]
            inside = b["seg"][j, 0] >= 0
            assert inside.all()                         # all toy cases are larger than the patch: nothing padded
            # locate the crop: its voxel values must reappear at one offset of the case
            found = False
            pos = np.argwhere(np.isclose(img[0], b["data"][j, 0, 0, 0, 0]))
            for (z, y, x) in pos:
                if z + ps[0] <= img.shape[1] and y + ps[1] <= img.shape[2] and x + ps[2] <= img.shape[3] and \
                        np.array_equal(img[0, z:z + ps[0], y:y + ps[1], x:x + ps[2]], b["data"][j, 0]):
                    assert np.array_equal(seg[z:z + ps[0], y:y + ps[1], x:x + ps[2]], b["seg"][j, 0])
                    found = True
                    break
            assert found
            if j >= 4:
                assert (b["seg"][j, 0] > 0).any()
    # ---- a case smaller than the patch is padded: image with zeros, segmentation with -1
    small_folder = str(tmp_path / "small")
    os.makedirs(small_folder)
    img, seg = _write_case(small_folder, "tiny", (10, 24, 12), 99)
    dl = DataLoader3D(load_dataset(small_folder), ps, ps, 2, False, oversample_foreground_percent=0.0, pad_mode="constant")
    b = next(dl)
    assert (b["seg"] == -1).sum() > 0 and np.all(b["data"][b["seg"] == -1] == 0)
    assert np.isclose((b["seg"][0, 0] >= 0).sum(), 10 * 16 * 12)
    # ---- deep-supervision targets: order-0 resize at pixel centres (voxel 2^i * o + 2^(i-1))
    s = np.arange(2 * 1 * 8 * 8 * 8, dtype=np.float32).reshape(2, 1, 8, 8, 8)
    t = downsample_seg_for_ds(s, 3)
    assert [x.shape for x in t] == [(2, 1, 8, 8, 8), (2, 1, 4, 4, 4), (2, 1, 2, 2, 2)]
    assert np.array_equal(t[1], s[:, :, 1::2, 1::2, 1::2]) and np.array_equal(t[2], s[:, :, 2::4, 2::4, 2::4])
    # ---- provider contract of the trainers
    plans = {"patch_size": ps, "batch_size": 2, "num_pool": 3, "base_num_features": 8, "num_classes": 3, "num_input_channels": 1}
    prov = PreprocessedDataProvider({"Task900_Toy": folder}, fold=0)
    d = next(prov("Task900_Toy", "train", plans))
    assert tuple(d["data"].shape) == (2, 1) + ps and [tuple(x.shape) for x in d["target"]] == [(2, 1, 16, 16, 16), (2, 1, 8, 8, 8), (2, 1, 4, 4, 4)]
    assert set(d["keys"]) <= set(tr) and float(d["target"][0].min()) >= 0
    assert set(next(prov("Task900_Toy", "val", plans))["keys"]) <= set(val)
    # ---- the augmentation hook (VERDICT r5 missing-3): a batchgenerators-style transform(**batch) -> batch sits where the reference
    # wraps its loaders (MH.py:904-922), BEFORE label -1 -> 0 and the deep-supervision down-sampling, training split only
    seen = {}

    def mirror_x(**b):
        seen["keys"], seen["min_seg"] = set(b), float(b["seg"].min())
        b["data"], b["seg"] = b["data"][..., ::-1].copy(), b["seg"][..., ::-1].copy()
        return b
    np.random.seed(3)
    plain = next(PreprocessedDataProvider({"Task900_Toy": folder}, fold=0)("Task900_Toy", "train", plans))
    np.random.seed(3)
    prov2 = PreprocessedDataProvider({"Task900_Toy": folder}, fold=0, train_transform=mirror_x)
    aug = next(prov2("Task900_Toy", "train", plans))
    assert seen["keys"] >= {"data", "seg", "keys", "properties"}
    assert torch.equal(aug["data"], plain["data"].flip(-1)) and torch.equal(aug["target"][0], plain["target"][0].flip(-1))
    # the down-sampled targets are taken from the TRANSFORMED segmentation
    assert torch.equal(aug["target"][1], torch.from_numpy(downsample_seg_for_ds(aug["target"][0].numpy(), 3)[1]))
    np.random.seed(4)
    v1 = next(prov2("Task900_Toy", "val", plans))
    np.random.seed(4)
    v2 = next(PreprocessedDataProvider({"Task900_Toy": folder}, fold=0)("Task900_Toy", "val", plans))
    assert torch.equal(v1["data"], v2["data"])                # no val_transform given: validation batches untouched


def test_evaluator_matches_the_reference_function_executed_on_recorded_volumes(golden_dir):
    """tests/golden/evaluator_reference.*: ``compute_scores_and_build_dict`` of nnunet_ext/evaluation/evaluator2.py:60-109 EXECUTED
    (oracle/make_goldens_evaluator.py) on seeded volumes -- random errors, a class absent from both volumes (None), a class that
    is only predicted (0.0), a perfect case, validation-only and validation + training case lists.  Both the oracle restatement and
    the product module must reproduce every number."""
    import json
    from oracle import evaluation as oev
    from lifelong_nnunet_amd.evaluation import compute_scores_and_build_dict
    arr = np.load(golden_dir + "/evaluator_reference.npz")
    meta = json.load(open(golden_dir + "/evaluator_reference.json"))
    K = meta["num_classes"]
    for key, names in (("validation_only", meta["splits"]["val"]), ("include_training_data", meta["splits"]["val"] + meta["splits"]["train"])):
        ref = meta["results"][key]
        assert list(ref) == names                                           # the reference's case order: val first, then train
        cases = {n: (arr["out::" + n], arr["tgt::" + n]) for n in names}
        got = compute_scores_and_build_dict(cases, K)
        assert list(got) == names
        for n in names:
            orc = oev.case_scores(*cases[n], K)
            assert list(got[n]) == list(ref[n]) == list(orc) == [f"mask_{c}" for c in range(1, K + 1)]
            for m in ref[n]:
                for metric in ("IoU", "Dice"):
                    r = ref[n][m][metric]
                    for val in (got[n][m][metric], orc[m][metric]):
                        assert (r is None and val is None) or (r is not None and val is not None and abs(val - r) <= 1e-12), (n, m, metric, r, val)
    assert meta["results"]["validation_only"]["case_b"]["mask_3"] == {"IoU": None, "Dice": None}
    assert meta["results"]["validation_only"]["case_c"]["mask_2"] == {"IoU": 0.0, "Dice": 0.0}
    assert meta["results"]["validation_only"]["case_d"]["mask_1"] == {"IoU": 1.0, "Dice": 1.0}


def test_evaluator_summary_matches_the_reference_arithmetic():
    """lifelong-nnunet_amd/evaluation.py against the reference's own scikit-learn call (evaluator2.py:88-107, restated in
    oracle/evaluation.py): random label volumes, a class absent from both volumes (-> None), a class only predicted."""
    from oracle import evaluation as oev
    from lifelong_nnunet_amd.evaluation import compute_scores_and_build_dict, summarize
    rng = np.random.RandomState(0)
    cases = {}
    for i in range(4):
        tgt = rng.randint(0, 3, (9, 11, 7))
        out = np.where(rng.rand(9, 11, 7) < 0.8, tgt, rng.randint(0, 3, (9, 11, 7)))
        cases[f"case_{i}"] = (out, tgt)
    bg = np.zeros((5, 5, 5), int)
    cases["all_background"] = (bg, bg)                                   # every mask None
    only_pred = bg.copy(); only_pred[1, 1, 1] = 2
    cases["false_positive_only"] = (only_pred, bg)                       # mask_2: IoU = Dice = 0, mask_1 None
    got = compute_scores_and_build_dict(cases, 2)
    assert list(got) == list(cases)
    for k, (out, tgt) in cases.items():
        exp = oev.case_scores(out, tgt, 2)
        assert list(got[k]) == ["mask_1", "mask_2"]
        for m in exp:
            for metric in ("IoU", "Dice"):
                a, b = got[k][m][metric], exp[m][metric]
                assert (a is None and b is None) or abs(a - b) <= 1e-12, (k, m, metric, a, b)
    assert got["all_background"]["mask_1"] == {"IoU": None, "Dice": None}
    assert got["false_positive_only"]["mask_2"] == {"IoU": 0.0, "Dice": 0.0} and got["false_positive_only"]["mask_1"]["Dice"] is None
    sm = summarize(got)
    assert sm["mask_1"]["Dice"]["n"] == 4 and sm["mask_2"]["Dice"]["n"] == 5
    assert abs(sm["mask_1"]["Dice"]["mean"] - np.mean([got[f"case_{i}"]["mask_1"]["Dice"] for i in range(4)])) < 1e-12


def test_load_checkpoint_written_by_the_reference_classes(golden_dir):
    """tests/golden/checkpoint_reference.* (oracle/make_goldens_checkpoint.py): ``state_dict()`` of the REFERENCE's
    ``MultiHead_Module`` (two heads, head B trained for three steps) and of the ``torch.optim.SGD`` that trained it -- what
    MH.py:1164-1197 / upstream ``save_checkpoint`` put into a ``.model`` file.  ``load_checkpoint_ram`` (MH.py:1278-1313)
    must rebuild both heads, the running model, the per-parameter momentum buffers and the epoch from it, and
    ``save_checkpoint`` must write the same state dict back, key for key and bit for bit."""
    import json
    from collections import OrderedDict
    from lifelong_nnunet_amd import get_trainer_class
    meta = json.load(open(os.path.join(golden_dir, "checkpoint_reference.json")))
    arr = np.load(os.path.join(golden_dir, "checkpoint_reference.npz"))
    sd = OrderedDict((k, torch.from_numpy(arr["sd::" + k])) for k in meta["state_dict_keys"])
    opt_state = {i: {"momentum_buffer": torch.from_numpy(arr[f"opt::{i}"])} for i in meta["optimizer_state_indices"]}
    group = dict(meta["optimizer_group"], params=meta["optimizer_params"], maximize=False, foreach=None, differentiable=False, fused=None)
    ckpt = {"epoch": meta["epoch"], "state_dict": OrderedDict(("module." + k, v) for k, v in sd.items()),      # DataParallel prefix, as upstream strips it
            "optimizer_state_dict": {"state": opt_state, "param_groups": [group]}, "lr_scheduler_state_dict": None,
            "plot_stuff": ([0.5, 0.4], [0.6, 0.5], [], []), "best_stuff": (None, None, None)}
    plans = {"patch_size": (16, 16, 16), "batch_size": 2, "num_pool": 2, "base_num_features": 8, "num_classes": 3, "num_input_channels": 1}
    tr = get_trainer_class("multihead")("seg_outputs", "taskA", plans=plans, device="cpu")
    tr.initialize(True, num_epochs=5)
    # the reference keeps the head list / active task in <ext>_trained_on.pkl (MH.py:1174-1180, 1283-1290)
    tr.already_trained_on = {"0": {"tasks_at_time_of_checkpoint": meta["heads"], "active_task_at_time_of_checkpoint": meta["active_task"]}}
    tr.load_checkpoint_ram(ckpt, train=True)
    assert list(tr.mh_network.heads.keys()) == meta["heads"] and str(tr.mh_network.active_task) == meta["active_task"]
    got = tr.mh_network.state_dict()
    assert list(got.keys()) == meta["state_dict_keys"]
    for k in meta["state_dict_keys"]:
        assert torch.equal(got[k].cpu(), sd[k]), k
    # the running model carries body + the ACTIVE head (B); head A is a different tensor set
    model = dict(tr.network.named_parameters())
    for n, p in model.items():
        assert torch.equal(p.detach().cpu(), sd["model." + n])
        if n.startswith("seg_outputs."):
            assert torch.equal(p.detach().cpu(), sd["heads.taskB." + n])
    assert any(not torch.equal(sd["heads.taskA." + n], sd["heads.taskB." + n]) for n in model if n.startswith("seg_outputs."))
    # momentum buffers: torch.optim.SGD indexes the TRAINABLE parameters in network.parameters() order
    names = [n for n, _ in tr.optimizer._trainable()]
    assert names == meta["optimizer_param_names"]
    by_name = dict(tr.network._named)
    for i in meta["optimizer_state_indices"]:
        s = by_name[names[i]]._lnn_slot
        assert torch.equal(tr.network.arena.momentum[s.offset:s.offset + s.numel].cpu(), opt_state[i]["momentum_buffer"].reshape(-1)), names[i]
    assert set(range(len(names))) - set(meta["optimizer_state_indices"]) == {names.index("seg_outputs.0.weight")}   # never stepped: no state
    assert tr.epoch == meta["epoch"] and tr.all_tr_losses == [0.5, 0.4] and tr.optimizer.param_groups[0]["lr"] == meta["optimizer_group"]["lr"]
    # ... and the way back: the checkpoint this trainer writes equals the reference's, tensor for tensor
    out = tr.save_checkpoint(None)
    assert list(out["state_dict"].keys()) == meta["state_dict_keys"]
    assert all(torch.equal(out["state_dict"][k], sd[k]) for k in sd)
    assert sorted(out["optimizer_state_dict"]["state"].keys()) == meta["optimizer_state_indices"]
    assert all(torch.equal(out["optimizer_state_dict"]["state"][i]["momentum_buffer"], opt_state[i]["momentum_buffer"])
               for i in meta["optimizer_state_indices"])


@pytest.mark.parametrize("pools,kernels", [(None, None),
                                           ([[1, 2, 2], [2, 2, 2]], [[1, 3, 3], [3, 3, 3], [3, 3, 3]])])
def test_backward_plan_invariants_of_the_fused_normalisation_reduce(pools, kernels):
    """The engine's plan (built on the CPU: no kernel runs) -- what the fused data-gradient + normalisation-reduce call of the backward
    relies on: the second block of every stage consumes exactly the first block's output (its gx IS that block's gz, written once, no
    concatenation), the pair is adjacent in the execution order (the reduce leaves its sums in the lane's workspace; nothing may run
    between the data gradient of block 1 and the apply pass of block 0), and every other block has no such partner."""
    net = Generic_UNet(2, 8, 3, 2, patch_size=(8, 16, 16), batch_size=2, device='cpu', pool_op_kernel_sizes=pools,
                       conv_kernel_sizes=kernels)
    eng = net.engine_for(torch.zeros(2, 2, 8, 16, 16))
    order = list(eng.order)
    partners = 0
    for blk in eng.blocks:
        xb = blk.x_block
        if xb is None:
            continue
        partners += 1
        assert blk.gx is xb.gz and blk.x is xb.z and blk.gx2 is None and blk.x2 is None and not blk.gx_accumulate
        assert tuple(blk.strides) == (1, 1, 1) and blk.cin == xb.cout and blk.in_dims == tuple(xb.y.shape[1:4])
        assert order.index(blk) == order.index(xb) + 1
    assert partners == len(eng.blocks) // 2
    firsts = [b for b in eng.blocks if b.x_block is None]
    assert all(not any(o.x_block is b for o in firsts) for b in firsts)


def test_bench_workload_plans_build_their_networks():
    """Every workload bench.py can be asked for names a plan the engine accepts (built on the CPU at a reduced in-plane size with the
    same poolings / kernels / channels): the anisotropic Prostate-shaped plan runs its [1,3,3] / [1,2,2] stages on the generic
    kernels and its lower stages on the isotropic ones, and its FLOP count scales to the 3.39 TFLOP per patch the bench reports."""
    import bench
    from lifelong_nnunet_amd.engine import ConvBlock
    assert set(bench.WORKLOADS) == {"c1", "c2", "c3", "c4", "c5", "prostate"}
    for name, (plans, ext, desc) in bench.WORKLOADS.items():
        q = 2 ** plans["num_pool"]
        pools = plans.get("pool_op_kernel_sizes")
        small = tuple(q for _ in range(3)) if pools is None else tuple(
            int(np.prod([p[a] for p in pools])) * (4 if a == 0 and name == "prostate" else 1) for a in range(3))
        net = Generic_UNet(plans["num_input_channels"], plans["base_num_features"], plans["num_classes"], plans["num_pool"],
                           patch_size=small, batch_size=1, device="cpu", pool_op_kernel_sizes=pools,
                           conv_kernel_sizes=plans.get("conv_kernel_sizes"))
        eng = net.engine_for(torch.zeros((1, plans["num_input_channels"]) + small))
        assert len(eng.segs) == plans["num_pool"] and ext in ("sequential", "ewc", "lwf", "rehearsal_ewc")
        if name == "prostate":
            blocks = [b for b in eng.order if isinstance(b, ConvBlock)]
            assert not blocks[0].iso and blocks[0].kernel == (1, 3, 3) and any(b.iso for b in blocks)
            assert eng.feats == [32, 64, 128, 256, 320, 320, 320]
            fl, _ = eng.flops_per_patch()
            full = fl * (20 * 320 * 256) / (small[0] * small[1] * small[2])
            assert abs(full / 1e9 - 3385.56) < 1.0


def test_committed_counter_passes_belong_to_the_library_the_tree_builds():
    """bench.py quotes HBM bytes per launch / shader clock / matrix-pipe occupancy from committed rocprofv3 counter passes and only for
    the binary they were collected with (``so_sha256``).  The build is reproducible, so the library built from this tree is normally that
    binary: the counter files that match its sha256 hold every key the bench line reads (all three families)."""
    import bench
    sha = bench.so_sha256()
    tj = bench.committed_pmc("pmc_traffic.json", sha)
    cj = bench.committed_pmc("pmc_mfma_clock.json", sha)
    if tj is None or cj is None:        # a library built by another toolchain: the bench line then carries traffic: null and says why
        pytest.skip(f"no profiles/rNN_pmc_*.json for liblnn_hip.so {sha[:16]} (counters are re-collected by tools/gpu_r5_final.sh)")
    for fam in ("fwd", "dgrad", "wgrad"):
        b = tj[1]["kernels"][fam]["hbm_bytes_per_launch_corrected"]
        assert 1e9 < b < 5e9, (fam, b)                         # dec4.0 launches: 1.9 GB algorithmic
    fams = {k: v for k, v in cj[1]["families"].items() if v}
    assert fams and all(1.0 < v["clock_ghz"] < 2.5 and 0.0 < v["mfma_busy_frac_in_cycles"] < 1.0 for v in fams.values())


def _rebuild_reference_side_files(root, golden_dir, method):
    """The files the reference's ``run_training`` tail wrote (oracle/make_goldens_sidedata.py), re-created under ``root`` from their
    recorded content: ``write_pickle`` = plain ``pickle.dump`` of {task: {name: tensor}}; ``<ext>_trained_on.pkl`` = the JSON text
    ``save_json`` writes, paths made absolute under ``root``.  Returns (meta, arrays)."""
    import pickle
    from collections import OrderedDict
    meta = json.load(open(f"{golden_dir}/sidedata_reference.json"))[method]
    arr = np.load(f"{golden_dir}/sidedata_reference.npz")
    for rel in meta["files"]:
        fname = os.path.basename(rel)
        per_task = OrderedDict()
        for key in arr.files:
            m_, f_, task, name = key.split("::")
            if m_ == method and f_ == fname:
                per_task.setdefault(task, OrderedDict())[name] = torch.from_numpy(arr[key].copy())
        os.makedirs(os.path.join(root, os.path.dirname(rel)), exist_ok=True)
        with open(os.path.join(root, rel), "wb") as f:
            pickle.dump(per_task, f)
    rec = json.loads(json.dumps(meta["already_trained_on"]))
    for k in ("fisher_at", "params_at", "scores_at"):
        if k in rec["0"]:
            rec["0"][k] = os.path.join(root, rec["0"][k])
    with open(os.path.join(root, meta["trained_on_file"]), "w") as f:
        json.dump(rec, f, indent=4, sort_keys=True)
    return meta, arr


TOY_PLANS = {"patch_size": (16, 16, 16), "batch_size": 2, "num_pool": 2, "base_num_features": 8, "num_classes": 3,
             "num_input_channels": 1, "synthetic_period": 4}


@pytest.mark.parametrize("method", ["ewc", "rw"])
def test_trainers_restore_the_side_data_the_reference_wrote(tmp_path, golden_dir, method):
    """SURVEY.md 8 row f4 / VERDICT r5 missing-1: ``fisher_values.pkl`` / ``param_values.pkl`` (/ ``score_values.pkl``) written by the
    REFERENCE's ``run_training`` tail (EWC.py:205-228, RW.py:267-300 -- executed by oracle/make_goldens_sidedata.py) are found through
    ``already_trained_on`` and loaded by the product's constructor (EWC.py:66-78, RW.py:77-84) and again by
    ``initialize(prev_trainer_path=...)`` (EWC.py:104-115); the loss object regularises with them.  The Fisher on disk is the
    ``ewc::fisherA`` of the trainer-flow fixture (same run of the reference)."""
    from lifelong_nnunet_amd import get_trainer_class
    meta, arr = _rebuild_reference_side_files(str(tmp_path), golden_dir, method)
    cls = get_trainer_class(method)
    rec = cls.read_trained_on_file(str(tmp_path / meta["trained_on_file"]))          # JSON under a .pkl name
    assert rec["0"]["finished_training_on"] == ["taskA"] and rec["0"]["fisher_at"].startswith(str(tmp_path))
    tr = cls("seg_outputs", "taskB", plans=dict(TOY_PLANS), device="cpu", already_trained_on=rec)
    dicts = {"fisher_values.pkl": tr.fisher, "param_values.pkl": tr.params}
    if method == "rw":
        dicts["score_values.pkl"] = tr.scores
    n = 0
    for key in arr.files:
        m_, f_, task, name = key.split("::")
        if m_ == method:
            assert torch.equal(dicts[f_][task][name].cpu(), torch.from_numpy(arr[key])), key
            n += 1
    assert n == sum(len(v) for d in dicts.values() for v in d.values()) and list(tr.fisher.keys()) == ["taskA"]
    if method == "ewc":
        ref = np.load(f"{golden_dir}/trainer_reference.npz")
        names = json.load(open(f"{golden_dir}/trainer_reference.json"))["ewc_flow"]["names"]
        flat = torch.cat([tr.fisher["taskA"][k].reshape(-1).float() for k in names]).numpy()
        assert np.array_equal(flat[::7], ref["ewc::fisherA::sub"])
        flat = torch.cat([tr.params["taskA"][k].reshape(-1).float() for k in names]).numpy()
        assert np.array_equal(flat[::7], ref["ewc::paramsA::sub"])
    # initialize(prev_trainer_path=...) re-reads them and hands them to the loss
    tr.fisher, tr.params = {}, {}
    tr.initialize(True, num_epochs=1, prev_trainer_path=str(tmp_path))
    assert list(tr.fisher.keys()) == ["taskA"] and tr.loss.tasks == (["taskA"] if method == "ewc" else tr.loss.tasks)
    assert tr.loss.fisher is tr.fisher and tr.loss.params is tr.params


@pytest.mark.parametrize("method", ["ewc", "rw"])
def test_side_data_round_trip_through_the_output_folder(tmp_path, method):
    """What the product writes at the end of a task (``save_fisher_and_params`` / ``save_f_p_s_values``): the reference's file names
    and dictionary layout (plain pickles of CPU tensors), the paths entered in ``already_trained_on`` ONCE, the ``<ext>_trained_on.pkl``
    file and the ``.pkl`` next to the final checkpoint updated -- and a trainer constructed from that record (position 12 of the stored
    constructor arguments, MH.py:1199-1208) starts with the same tensors."""
    import pickle
    from lifelong_nnunet_amd import get_trainer_class
    cls = get_trainer_class(method)
    out = tmp_path / "results" / "TaskA_TaskB" / "fold_0"
    tr = cls("seg_outputs", "taskA", plans=dict(TOY_PLANS), device="cpu", output_folder=str(out))
    assert tr.trained_on_path == str(tmp_path / "results")
    g = torch.Generator().manual_seed(5)
    names = ["conv_blocks_context.0.blocks.0.conv.weight", "seg_outputs.1.weight"]
    mk = lambda: {"taskA": {k: torch.randn((3, 4), generator=g) for k in names}}
    tr.fisher, tr.params = mk(), mk()
    tr.fisher["taskA"]["seg_outputs.0.weight"] = torch.tensor([1.0])                 # the grad-less head's Fisher (EWC.py:300-301)
    if method == "rw":
        tr.scores = mk()
        tr.save_f_p_s_values()
        sub, keys = "rw_data", ("fisher_at", "params_at", "scores_at")
    else:
        tr.save_fisher_and_params()
        sub, keys = "ewc_data", ("fisher_at", "params_at")
    fold = tr.already_trained_on["0"]
    assert fold["fisher_at"] == str(tmp_path / "results" / sub / "fisher_values.pkl") and all(os.path.isfile(fold[k]) for k in keys)
    disk = pickle.load(open(fold["fisher_at"], "rb"))
    assert list(disk.keys()) == ["taskA"] and all(v.device.type == "cpu" for v in disk["taskA"].values())
    rec = cls.read_trained_on_file(str(tmp_path / "results" / f"{method}_trained_on.pkl"))
    assert rec["0"]["fisher_at"] == fold["fisher_at"]
    info = pickle.load(open(str(out / "model_final_checkpoint.model.pkl"), "rb"))
    assert info["init"][12]["0"]["params_at"] == fold["params_at"] and info["name"] == cls.__name__
    # a second task overwrites the files but not the record
    tr.fisher["taskB"] = {k: torch.zeros(2) for k in names}
    before = dict(fold)
    (tr.save_f_p_s_values if method == "rw" else tr.save_fisher_and_params)()
    assert {k: fold[k] for k in keys} == {k: before[k] for k in keys}
    assert list(pickle.load(open(fold["fisher_at"], "rb")).keys()) == ["taskA", "taskB"]
    # restore
    tr2 = cls("seg_outputs", "taskC", plans=dict(TOY_PLANS), device="cpu", output_folder=str(out), already_trained_on=info["init"][12])
    assert list(tr2.fisher.keys()) == ["taskA", "taskB"]         # the record points at the files, and those hold both tasks now
    tr3 = cls("seg_outputs", "taskC", plans=dict(TOY_PLANS), device="cpu", output_folder=str(out), already_trained_on=rec)
    for k, v in tr.fisher["taskA"].items():
        assert torch.equal(tr3.fisher["taskA"][k], v)
    # without an output folder nothing is written and nothing is recorded
    tr4 = cls("seg_outputs", "taskA", plans=dict(TOY_PLANS), device="cpu")
    tr4.fisher, tr4.params = mk(), mk()
    (tr4.save_f_p_s_values if method == "rw" else tr4.save_fisher_and_params)()
    assert tr4.already_trained_on["0"]["fisher_at"] is None


def test_lwf_restore_record(tmp_path):
    """LWF.py:60-87: the freeze-run record of the LwF trainer -- defaults, ``freeze_run`` derived from it, reset per task."""
    from lifelong_nnunet_amd import get_trainer_class
    cls = get_trainer_class("lwf")
    tr = cls("seg_outputs", "taskA", plans=dict(TOY_PLANS), device="cpu")
    fold = tr.already_trained_on["0"]
    assert fold["freeze_run_finished"] is False and fold["freezed_model_at"] is None and fold["ftasks_at_time_of_checkpoint"] == []
    assert fold["used_lwf_temperature"] == 2.0 and tr.freeze_run is True
    rec = {"0": dict(fold, freeze_run_finished=True, freezed_model_at="/x/model_freezed.model",
                     ftasks_at_time_of_checkpoint=["taskA", "taskB"], factive_task_at_time_of_checkpoint="taskB")}
    tr2 = cls("seg_outputs", "taskB", plans=dict(TOY_PLANS), device="cpu", already_trained_on=rec)
    assert tr2.freeze_run is False
    tr2._reset_restore_record()
    assert rec["0"]["freeze_run_finished"] is False and rec["0"]["freezed_model_at"] is None


def test_deferred_loss_behaves_like_the_numpy_scalar_the_reference_returns():
    """``run_iteration`` returns a DeferredLoss inside the epoch loop (MH.py:655 returns a numpy scalar there); an override
    that compares it, does arithmetic on it or asks numpy about it must keep working."""
    from lifelong_nnunet_amd.optim import DeferredLoss

    class Ctrl:
        def get(self):
            return (4.0, 0.0, 0.75)
    d = DeferredLoss(Ctrl())
    assert d > 0.5 and d < 1 and d >= 0.75 and d <= 0.75 and d == 0.75 and d != 0.5
    assert d + 1 == 1.75 and 1 + d == 1.75 and 2 * d == 1.5 and d * 2 == 1.5 and d - 0.25 == 0.5 and 1 - d == 0.25 and d / 3 == np.float32(0.75) / 3
    assert -d == -0.75 and abs(-1 * d) == 0.75 and d ** 2 == 0.5625 and bool(d) and not np.isnan(d) and np.isfinite(d)
    assert float(d) == 0.75 and f"{d:.2f}" == "0.75" and np.mean([d, DeferredLoss(Ctrl())]) == 0.75
    assert DeferredLoss(Ctrl()) + DeferredLoss(Ctrl()) == 1.5 and max(d, 0.5) is d


def test_nested_resplit_quirk_is_announced_once():
    """reproducing the reference's nested re-split silently turns every tensor into body -- say so, once."""
    import warnings as w
    MultiHead_Module._warned_nested_resplit = False
    mh = MultiHead_Module(Generic_UNet, "tu.1", "taskA", None, 1, 8, 3, 2, device="cpu")
    with w.catch_warnings(record=True) as rec:
        w.simplefilter("always")
        mh.update_after_iteration()
        mh.update_after_iteration()
    msgs = [str(r.message) for r in rec if "nested re-split" in str(r.message)]
    assert len(msgs) == 1 and "reference_nested_resplit = False" in msgs[0]


def test_every_environment_switch_is_documented():
    """VERDICT r5 task 10: every ``LNN_*`` variable the product reads is named in INTEGRATION.md's switch table (with its shipped
    default), no stale names remain there, and none is set while the test suite runs -- the suite exercises the shipped paths."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    found = set()
    for base, exts in ((os.path.join(root, "lifelong-nnunet_amd"), (".py", ".hip", ".h")), (root, ("bench.py",))):
        for dp, _, fns in os.walk(base):
            if base == root and dp != root:
                continue
            for fn in fns:
                if fn.endswith(exts):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    found |= set(re.findall(r'getenv\("(LNN_[A-Z0-9_]+)"', txt))
                    found |= set(re.findall(r'environ(?:\.get)?[\[(]\s*"(LNN_[A-Z0-9_]+)"', txt))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    table = doc[doc.index("| switch | read by |"):doc.index("test_every_environment_switch_is_documented")]
    named = set(re.findall(r"`(LNN_[A-Z0-9_]+)`", table))
    assert found, "the scan found nothing: the patterns no longer match the sources"
    assert found - named == set(), f"undocumented switches: {sorted(found - named)}"
    assert named - found == set(), f"documented but no longer read: {sorted(named - found)}"
    leaked = {k for k in os.environ if k.startswith("LNN_") and k not in ("LNN_RELERR_LOG",)}
    assert leaked <= {"LNN_FORCE_DP"}, f"switches set in the test environment: {sorted(leaked)}"


def test_heads_and_body_are_module_trees_of_the_network_classes():
    """VERDICT r5 missing-2 (MHM.py:159-324 builds ``nn.Module`` bodies / heads): what ``get_heads()[task].children()`` /
    ``get_body().children()`` yield are the network's own module classes, state-dict keys unchanged, and ``replace_layers``
    (MHM.py:544-572) finds the modules of a head the way it finds them in the reference's."""
    from torch import nn
    from lifelong_nnunet_amd.network import ConvDropoutNormNonlin, StackedConvLayers
    mh = MultiHead_Module(Generic_UNet, "seg_outputs", "taskA", None, 1, 8, 3, 2, device="cpu")
    head = mh.get_heads()["taskA"]
    kids = dict(head.named_children())
    assert list(kids) == ["seg_outputs"] and isinstance(kids["seg_outputs"], nn.ModuleList) and len(kids["seg_outputs"]) == 2
    assert [n for n, _ in head.named_parameters()] == ["seg_outputs.0.weight", "seg_outputs.1.weight"]
    body = mh.get_body()
    assert set(n for n, _ in body.named_children()) == {"conv_blocks_localization", "conv_blocks_context", "tu"}
    first = body.conv_blocks_context[0]
    assert isinstance(first, StackedConvLayers) and first.input_channels == 1 and isinstance(first.blocks, nn.Sequential)
    blk = first.blocks[0]
    assert isinstance(blk, ConvDropoutNormNonlin) and isinstance(blk.lrelu, nn.LeakyReLU) and blk.lrelu.negative_slope == 1e-2
    assert blk.conv.weight.shape == (8, 1, 3, 3, 3) and blk.instnorm.weight.shape == (8,)
    # a deeper split: the head holds the whole encoder (+ tu + seg_outputs); replace_layers swaps every LeakyReLU in it
    mh2 = MultiHead_Module(Generic_UNet, "conv_blocks_context", "taskA", None, 1, 8, 3, 2, device="cpu")
    h2 = mh2.get_heads()["taskA"]
    n_lrelu = sum(isinstance(m, nn.LeakyReLU) for m in h2.modules())
    assert n_lrelu == 6                                        # 3 encoder stages x 2 blocks
    h2 = mh2.replace_layers(h2, nn.LeakyReLU, nn.ReLU())
    # (the ONE instance handed in sits at all six places, as in the reference: setattr(model, name, new))
    assert sum(isinstance(m, nn.LeakyReLU) for m in h2.modules()) == 0
    assert sum(isinstance(m, nn.ReLU) for _, m in h2.named_modules(remove_duplicate=False)) == 6
    assert list(mh2.heads["taskA"].state_dict().keys()) == list(h2.state_dict().keys())
