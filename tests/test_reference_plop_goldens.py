"""CPU tests against ``tests/golden/plop_reference.{json,npz}`` -- outputs of the REFERENCE's own PLOP / POD code
(embeddings.local_POD, MultipleOutputLossPLOP / MultipleOutputLossPOD, nnUNetTrainerPLOP / nnUNetTrainerPOD), executed
verbatim by ``oracle/make_goldens_plop.py``.  The oracle restatement (oracle/plop.py) is pinned here; the HIP path is
checked against the same fixtures in tests/test_plop_gpu.py."""
import json
import math
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import losses as olosses, plop as oplop, train as otrain
from oracle.unet import OracleGenericUNet
from lifelong_nnunet_amd.synthetic import make_patch_batch

BOOST = 60.0


@pytest.fixture(scope="module")
def ref(golden_dir):
    return (json.load(open(golden_dir + "/plop_reference.json")), np.load(golden_dir + "/plop_reference.npz"),
            np.load(golden_dir + "/trainer_reference.npz"))


class Counting:
    def __init__(self, items):
        self.items, self.n = items, 0

    def __iter__(self):
        return self

    def __next__(self):
        b = self.items[self.n % len(self.items)]
        self.n += 1
        return b


def ref_batches(task_seed, n):
    out = []
    for i in range(n):
        data, tgts = make_patch_batch(2, (16, 16, 16), 2, seed=task_seed + i)
        out.append({"data": data, "target": tgts, "keys": [f"case_{task_seed + i}_{b}" for b in range(2)]})
    return out


def test_local_pod_matches_reference(ref):
    meta, arr, _ = ref
    for i, c in enumerate(meta["local_pod"]["cases"]):
        a, b = torch.from_numpy(arr[f"pod::a{i}"]), torch.from_numpy(arr[f"pod::b{i}"])
        assert tuple(a.shape) == tuple(c["shape"])
        v = oplop.local_pod(a, b, c["scales"])
        assert abs(v - c["value"]) <= 1e-6 * abs(c["value"]), (c, v)
        assert oplop.local_pod(a, a, c["scales"]) == 0.0
    assert meta["local_pod"]["fails"] == {"non_square": "RuntimeError", "too_many_scales": "AssertionError"}
    with pytest.raises(RuntimeError):
        oplop.local_pod(torch.zeros(2, 2, 2, 8, 6), torch.zeros(2, 2, 2, 8, 6), 3)
    with pytest.raises(AssertionError):
        oplop.local_pod(torch.zeros(2, 2, 2, 2, 2), torch.zeros(2, 2, 2, 2, 2), 3)


def _loss_inputs(meta, arr):
    m = meta["plop_loss"]
    x = [torch.from_numpy(arr[f"loss::x{i}"]).clone().requires_grad_(True) for i in range(3)]
    x_o = [torch.from_numpy(arr[f"loss::xo{i}"]) for i in range(3)]
    y = [torch.from_numpy(arr[f"loss::y{i}"]) for i in range(3)]
    thr = {i: torch.tensor(t) for i, t in enumerate(m["thresholds"])}
    interm = OrderedDict((k, torch.from_numpy(arr[f"loss::h_{k}"])) for k in m["layers"])
    old = OrderedDict((k, torch.from_numpy(arr[f"loss::ho_{k}"])) for k in m["layers"])
    return m, x, x_o, y, thr, interm, old


def test_plop_and_pod_loss_match_reference(ref):
    meta, arr, _ = ref
    m, x, x_o, y, thr, interm, old = _loss_inputs(meta, arr)
    w = np.asarray(m["weights"])
    for idx, v in m["per_level"]:
        got = float(oplop.pseudo_label_loss(x[idx], x_o[idx], y[idx], thr[idx], math.log(3)))
        assert abs(got - v) <= 1e-6 * abs(v), (idx, got, v)
    assert [i for i, _ in m["per_level"]] == [0, 1]                       # the zero-weight level is skipped
    d = oplop.dist_term(interm, old, m["pod_lambda"], m["scales"])
    assert abs(d - m["dist"]) <= 1e-4 * abs(m["dist"])                      # the fixture's dist is a float32 difference
    val = oplop.plop_loss(x, x_o, y, w, interm, old, thr, math.log(3), m["pod_lambda"], m["scales"])
    assert abs(float(val) - m["value"]) <= 1e-6 * abs(m["value"])
    val.backward()
    for i in range(2):
        exp = arr[f"loss::dx{i}"]
        assert np.linalg.norm(x[i].grad.numpy() - exp) <= 1e-6 * np.linalg.norm(exp)
    assert x[2].grad is None
    base = float(olosses.multiple_output_loss([t.detach() for t in x], y, w))
    assert abs(base - meta["pod_loss"]["base"]) <= 1e-6 * abs(base)
    assert abs(oplop.pod_loss(base, interm, old, m["pod_lambda"], m["scales"]) - meta["pod_loss"]["value"]) <= 1e-6 * meta["pod_loss"]["value"]


EXEC_ORDER = ['conv_blocks_context.0.blocks.0.conv', 'conv_blocks_context.0.blocks.1.conv', 'conv_blocks_context.1.blocks.0.conv',
              'conv_blocks_context.1.blocks.1.conv', 'conv_blocks_context.2.0.blocks.0.conv', 'conv_blocks_context.2.1.blocks.0.conv',
              'tu.0', 'conv_blocks_localization.0.0.blocks.0.conv', 'conv_blocks_localization.0.1.blocks.0.conv', 'seg_outputs.0',
              'tu.1', 'conv_blocks_localization.1.0.blocks.0.conv', 'conv_blocks_localization.1.1.blocks.0.conv', 'seg_outputs.1']


def _oracle_flow(f, init, pod_trainer):
    """The flow of oracle/make_goldens_plop.py:flow on the oracle network; returns per-task losses, per-task per-layer POD
    values and the generator positions."""
    net = OracleGenericUNet(1, 8, 3, 2)
    net.load_state_dict(init)
    init_head = {k: v.clone() for k, v in init.items() if k.startswith("seg_outputs.")}
    if f["boost"] != 1.0:
        with torch.no_grad():
            for n, p in net.named_parameters():
                if n.startswith("seg_outputs."):
                    p.mul_(f["boost"])
    opt = otrain.make_optimizer(net)
    w = olosses.ds_loss_weights(2)
    gens = {t: Counting(ref_batches(f["seeds"][t], 8)) for t in f["tasks"]}
    taps = oplop.Taps(net)
    out = {}
    prev = None
    for ti, t in enumerate(f["tasks"]):
        losses, pods = [], []
        if ti == 0:
            for _ in range(2):
                b = next(gens[t])
                losses.append(otrain.run_iteration(net, opt, b["data"], b["target"], w)[0])
                taps.take()
        else:
            old = OracleGenericUNet(1, 8, 3, 2)
            old.load_state_dict(net.state_dict())
            old.eval()
            taps_old = oplop.Taps(old)
            thr, max_e = None, None
            if not pod_trainer:
                thr, max_e = oplop.thresholds(3, 2)
                for _ in range(2):
                    next(gens[prev])           # the extraction draws from the generator of the task trained before
            with torch.no_grad():
                for n, p in net.named_parameters():
                    if n.startswith("seg_outputs."):
                        p.copy_(init_head[n])  # add_new_task(use_init) + assemble_model
            base = (lambda o, y: olosses.multiple_output_loss(o, y, w)) if pod_trainer else None
            for _ in range(2):
                b = next(gens[t])
                # per-layer values, for the comparison with the recorded local_POD calls
                with torch.no_grad():
                    net(b["data"]); cur = taps.take()
                    old(b["data"]); o_ = taps_old.take()
                if ti >= 2:
                    cur = o_
                assert list(o_.keys()) == EXEC_ORDER
                pods += [oplop.local_pod(cur[k], o_[k], 3) for k in o_]
                losses.append(oplop.plop_iteration(net, old, taps, taps_old, opt, b["data"], b["target"], w, thr, max_e, 0.01, 3,
                                                   pod_only=pod_trainer, base_loss=base, alias_old=ti >= 2))
        out[t] = (losses, pods, {k: g.n for k, g in gens.items()})
        prev = t
    return out, net


def _same(a, b, rtol=1e-5):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    assert np.array_equal(np.isnan(a), np.isnan(b)), (a, b)
    m = ~np.isnan(a)
    assert np.all(np.abs(a[m] - b[m]) <= rtol * np.abs(b[m]) + 1e-9), (a, b)


def _theta_close(arr, key, net, names, rtol):
    flat = torch.cat([dict(net.named_parameters())[n].detach().reshape(-1) for n in names]).numpy()
    exp = arr[key + "::sub"]
    assert np.linalg.norm(flat[::7] - exp) <= rtol * np.linalg.norm(exp)


@pytest.mark.parametrize("which", ["plop_flow", "pod_flow"])
def test_oracle_flows_equal_reference(ref, which):
    meta, arr, tarr = ref
    f = meta[which]
    init = {n[6:]: torch.from_numpy(tarr[n]) for n in tarr.files if n.startswith("init::")}
    out, net = _oracle_flow(f, init, which == "pod_flow")
    for t in f["tasks"]:
        losses, pods, consumed = out[t]
        _same(losses, f["losses_" + t])
        _same(pods, f["pods_" + t], rtol=2e-4)
        assert consumed == f["consumed_after_" + t]
        for o in f["order_" + t]:
            assert o == EXEC_ORDER
    if which == "plop_flow":
        assert np.all(np.isnan(f["losses_taskC"]))                      # no confident voxel in the third task: CE over nothing
        e = f["extracted"]
        assert len(e) == 2 and all(abs(x["max_entropy"] - math.log(3)) < 1e-6 for x in e)
        assert all(np.allclose(v, 0.001) for x in e for v in x["thresholds"].values())
        assert f["thresholds_reset"] == [True, True]
        assert f["consumed_after_taskB"]["taskA"] == 4 and f["consumed_after_taskC"]["taskB"] == 4
    else:
        _theta_close(arr, "pod::final_theta", net, f["names"], 1e-5)
    assert not any(f["pods_taskC"]) and any(f["pods_taskB"])             # third task: hook aliasing -> POD is exactly 0
    assert f["hook_counts"] == {"network": 1, "network_old": 2}


def test_unconfident_plop_is_nan_in_the_reference(ref):
    meta, arr, tarr = ref
    f = meta["plop_flow_unconfident"]
    assert np.all(np.isnan(f["losses_taskB"])) and not np.any(np.isnan(f["losses_taskA"]))
    init = {n[6:]: torch.from_numpy(tarr[n]) for n in tarr.files if n.startswith("init::")}
    g = dict(meta["plop_flow"]); g["boost"] = 1.0; g["tasks"] = ["taskA", "taskB"]
    out, _ = _oracle_flow(g, init, False)
    _same(out["taskA"][0], f["losses_taskA"])
    assert np.all(np.isnan(out["taskB"][0]))
    _same(out["taskB"][1], f["pods_taskB"], rtol=2e-4)
