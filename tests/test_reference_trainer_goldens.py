"""CPU tests against ``tests/golden/trainer_reference.{json,npz}`` -- outputs of the REFERENCE's own trainer methods,
executed verbatim by ``oracle/make_goldens_trainers.py`` (MH.py / EWC.py / RW.py / REH.py, see its docstring for the
file:line list).  Both the oracle restatement and the product's host logic are checked here; the HIP trainers are
checked against the same fixtures in tests/test_trainer_goldens_gpu.py."""
import json
import random
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import losses as olosses, train as otrain
from oracle.unet import OracleGenericUNet
from lifelong_nnunet_amd import get_trainer_class
from lifelong_nnunet_amd.dataloading import do_split
from lifelong_nnunet_amd.losses import ds_loss_weights
from lifelong_nnunet_amd.network import Generic_UNet
from lifelong_nnunet_amd.synthetic import make_patch_batch


@pytest.fixture(scope="module")
def ref(golden_dir):
    return json.load(open(golden_dir + "/trainer_reference.json")), np.load(golden_dir + "/trainer_reference.npz")


def ref_batches(task_seed, n, patch=(16, 16, 16), B=2, num_pool=2):
    """The batches oracle/make_goldens_trainers.py:batches fed to the reference (same generator, same seeds)."""
    out = []
    for i in range(n):
        data, tgts = make_patch_batch(B, patch, num_pool, seed=task_seed + i)
        out.append({"data": data, "target": tgts, "keys": [f"case_{task_seed + i}_{b}" for b in range(B)]})
    return out


# ------------------------------------------------------------------------------------------------ structure
def _expected_channels(in_ch, base, ncls, num_pool, conv_upsampling, max_feat=320):
    """Channel layout of upstream Generic_UNet.  ``conv_upsampling=False`` (class default, what the reference's
    module-tree dump was printed with): the LAST conv of a stage that feeds a parameter-free Upsample must already reduce
    to the next skip's width; ``True`` (nnUNetTrainerV2's configuration, nnViTUNetTrainer.py:117-122): the transposed conv
    changes the width and both decoder convs keep the skip's width."""
    feats = [min(base * 2 ** d, max_feat) for d in range(num_pool + 1)]
    exp = {}
    cin = in_ch
    for d in range(num_pool):
        exp[f"conv_blocks_context.{d}.blocks.0.conv"] = (cin, feats[d])
        exp[f"conv_blocks_context.{d}.blocks.1.conv"] = (feats[d], feats[d])
        cin = feats[d]
    bott_out = feats[num_pool] if conv_upsampling else feats[num_pool - 1]
    exp[f"conv_blocks_context.{num_pool}.0.blocks.0.conv"] = (cin, feats[num_pool])
    exp[f"conv_blocks_context.{num_pool}.1.blocks.0.conv"] = (feats[num_pool], bott_out)
    for u in range(num_pool):
        skip = feats[num_pool - 1 - u]
        final = skip if (conv_upsampling or u == num_pool - 1) else feats[num_pool - 2 - u]
        exp[f"conv_blocks_localization.{u}.0.blocks.0.conv"] = (2 * skip, skip)
        exp[f"conv_blocks_localization.{u}.1.blocks.0.conv"] = (skip, final)
        exp[f"seg_outputs.{u}"] = (final, ncls)
    return exp


def test_module_tree_matches_the_reference_dump(ref):
    """test/network_architecture/test_MultiHead_Module.py:281-433 prints Generic_UNet(3, 5, 2, 3) with the class defaults.
    (1) every leaf path of that dump exists in the oracle AND the product network built with the same arguments
    (names + nesting are the API: Fisher dictionaries, split paths, freezing); (2) the dump's channel counts equal the
    upstream rule for convolutional_upsampling=False; (3) both networks equal the same rule for =True, the trainer's
    configuration -- the two layouts differ in exactly the 8 leaves the flag governs."""
    meta, _ = ref
    leaves = meta["module_tree"]["leaves"]
    ctor = meta["module_tree"]["ctor"]
    exp_false, exp_true = _expected_channels(*ctor, False), _expected_channels(*ctor, True)
    dump = {p: (cin, cout) for p, t, cin, cout in leaves if t.startswith("Conv")}
    assert dump == exp_false
    assert sum(exp_false[k] != exp_true[k] for k in exp_true) == 2 * (ctor[3] - 1) + 1      # + their 3 norm layers = 8 leaves
    for make in (lambda: OracleGenericUNet(*ctor), lambda: Generic_UNet(*ctor, device="cpu")):
        mods = dict(make().named_modules())
        for path, typ, cin, cout in leaves:
            assert path in mods, path
            w = mods[path].weight
            if typ.startswith("Conv"):
                assert tuple(w.shape[:2]) == (exp_true[path][1], exp_true[path][0]), path
            else:      # the norm layer that follows conv <path minus '.instnorm'>
                assert w.shape[0] == exp_true[path.replace(".instnorm", ".conv")][1], path
        # nothing with parameters beyond the dump's leaves except the transposed convs (the dump's `tu` are Upsample modules)
        extra = [n for n, m in mods.items() if getattr(m, "weight", None) is not None and n not in {l[0] for l in leaves}]
        assert all(n.startswith("tu.") for n in extra), extra


def test_ds_weights_and_reorder(ref):
    meta, _ = ref
    for k, w in meta["ds_weights"].items():
        assert np.allclose(ds_loss_weights(int(k)), w, rtol=0, atol=1e-15)
        assert np.allclose(olosses.ds_loss_weights(int(k)), w, rtol=0, atol=1e-15)
    r = meta["reorder"]
    tr = get_trainer_class("multihead")("seg_outputs", "t", device="cpu")
    tr.network = Generic_UNet(*r["ctor"], device="cpu")
    assert [n for n, _ in tr.network.named_parameters()] == r["before"]
    tr.reorder_UNet_components()
    assert [n for n, _ in tr.network.named_parameters()] == r["after"]


def test_do_split_matches_reference(ref):
    meta, _ = ref
    ds = OrderedDict((k, {"case": k}) for k in meta["do_split"]["keys"])
    for fold, exp in meta["do_split"]["folds"].items():
        tr, val = do_split(ds, int(fold))
        assert list(tr.keys()) == exp["train"] and list(val.keys()) == exp["val"], fold


def test_rehearsal_generators_match_reference(ref):
    """REH.py:65-173 executed by the reference on three fake task datasets: fused training keys (current task's split
    first, then the seeded samples of each previous head in draw order) and validation keys."""
    meta, _ = ref
    r = meta["rehearsal"]
    datasets = {k.split("/")[-1]: v for k, v in r["datasets"].items()}

    class Provider:
        def dataset_for(self, task):
            return OrderedDict((k, {"case": k}) for k in datasets[task])

        def generator_for(self, dataset, plans, split):
            return list(dataset.keys())

    for ext in ("rehearsal", "rehearsal_ewc"):
        tr = get_trainer_class(ext)("seg_outputs", r["current"], device="cpu", data_provider=Provider(),
                                    samples_in_perc=r["samples"], seed=r["seed"])

        class _MH:
            heads = OrderedDict((t, None) for t in r["heads"])
        tr.mh_network = _MH()
        state = random.getstate()
        tr_keys, val_keys = tr.get_basic_generators()
        assert tr_keys == r["train_keys_fused"] and val_keys == r["val_keys"]
        assert random.getstate() != state          # REH.py:169 re-seeds from the OS
    # the oracle's sampler restates the same draw
    prev_tr = [list(do_split(OrderedDict((k, 0) for k in datasets[t]), 0)[0].keys()) for t in r["heads"]]
    picked = otrain.rehearsal_sample(prev_tr, r["samples"], r["seed"])
    n_cur = len(do_split(OrderedDict((k, 0) for k in datasets[r["current"]]), 0)[0])
    assert sum(picked, []) == r["train_keys_fused"][n_cur:]


# ------------------------------------------------------------------------------------------------ online evaluation
def test_online_evaluation_per_subject(ref):
    meta, arr = ref
    ev = meta["online_eval"]
    exp = ev["validation_results"]["epoch_%d" % ev["epoch"]][ev["task"]]
    tps, fps, fns = [], [], []
    for bi in range(3):
        tp, fp, fn = olosses.online_dice_counts(torch.from_numpy(arr[f"eval::logits_{bi}"]), torch.from_numpy(arr[f"eval::target_{bi}"]))
        for got, key in ((tp, "tp"), (fp, "fp"), (fn, "fn")):
            assert np.array_equal(got.numpy(), arr[f"eval::{key}_{bi}"])
        tps.append(tp.numpy()); fps.append(fp.numpy()); fns.append(fn.numpy())
    # oracle restatement
    got = otrain.per_subject_dice(tps, fps, fns, ev["names_per_batch"])
    _same_results(got, exp)
    # product host logic (MH.py:963-1049)
    tr = get_trainer_class("multihead")("seg_outputs", ev["task"], device="cpu")
    tr.online_eval_tp, tr.online_eval_fp, tr.online_eval_fn = list(tps), list(fps), list(fns)
    tr.subject_names_raw = [np.array(n) for n in ev["names_per_batch"]]
    tr.epoch = ev["epoch"]
    summary = tr.finish_online_evaluation_extended(ev["task"])
    _same_results(tr.validation_results["epoch_%d" % ev["epoch"]][ev["task"]], exp)
    assert tr.online_eval_tp == [] and tr.subject_names_raw == []
    dices = [v["Dice"] for s in exp.values() for v in s.values()]
    assert abs(summary["mean_dice"] - np.nanmean(dices)) < 1e-12


def _same_results(got, exp):
    assert sorted(got.keys()) == sorted(exp.keys())
    for s in exp:
        assert sorted(got[s].keys()) == sorted(exp[s].keys())
        for m in exp[s]:
            for k in ("IoU", "Dice"):
                a, b = float(got[s][m][k]), float(exp[s][m][k])
                assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-6 * max(1.0, abs(b)), (s, m, k, a, b)


# ------------------------------------------------------------------------------------------------ oracle trainer flows
def _check(arr, key, d, names, sub, rtol):
    flat = torch.cat([d[n].detach().float().reshape(-1) for n in names]).numpy()
    exp = arr[key + "::sub"]
    got = flat[::sub]
    assert got.shape == exp.shape, key
    den = np.linalg.norm(exp) + 1e-30
    assert np.linalg.norm(got - exp) / den <= rtol, (key, np.linalg.norm(got - exp) / den)
    stats = arr[key + "::stats"]
    for i, n in enumerate(names):
        assert abs(float(d[n].double().norm()) - stats[i, 1]) <= rtol * max(stats[i, 1], 1e-12) + 1e-12, (key, n)


def _fresh_head(net, arr):
    """add_new_task(task, use_init=True) + assemble_model (MHM.py:435-458,326-377): head tensors <- the initial head state."""
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.startswith("seg_outputs."):
                p.copy_(torch.from_numpy(arr["init::" + n]))


def test_oracle_ewc_flow_equals_reference(ref):
    """EWC.py:179-310 + MH.py:598-656 executed by the reference == oracle.train (losses of both tasks, Fisher / theta* of both)."""
    meta, arr = ref
    e = meta["ewc_flow"]
    names = e["names"]
    net = OracleGenericUNet(1, 8, 3, 2)
    net.load_state_dict({n[6:]: torch.from_numpy(arr[n]) for n in arr.files if n.startswith("init::")})
    opt = otrain.make_optimizer(net)
    w = olosses.ds_loss_weights(2)
    bA = ref_batches(1000, 6)
    lA = [otrain.run_iteration(net, opt, b["data"], b["target"], w)[0] for b in bA[:3]]
    assert np.allclose(lA, e["lossesA"], rtol=1e-6)
    fA, pA = otrain.ewc_after_train(net, opt, [(b["data"], b["target"]) for b in bA[3:]], w)
    _check(arr, "ewc::fisherA", fA, names, 7, 1e-6)
    _check(arr, "ewc::paramsA", pA, names, 7, 1e-7)
    # task B: MH.py:551-566 registers a NEW head initialised from the very first head state (use_init) and assembles it;
    # penalty of task A with a fresh named_parameters() generator per iteration (EWC.py:247)
    _fresh_head(net, arr)
    fisher, params = {"taskA": fA}, {"taskA": pA}
    pen = lambda: olosses.ewc_penalty(net.named_parameters(), fisher, params, 0.4)
    bB = ref_batches(2000, 6)
    lB = [otrain.run_iteration(net, opt, b["data"], b["target"], w, extra_loss=pen)[0] for b in bB[:3]]
    assert np.allclose(lB, e["lossesB"], rtol=1e-6), (lB, e["lossesB"])
    # after_train of task B: the loss object still holds the generator handed over after the last iteration -> the penalty
    # is part of the FIRST after_train batch only and absent from the last one, whose gradient becomes the Fisher
    fB, pB = otrain.ewc_after_train(net, opt, [(b["data"], b["target"]) for b in bB[3:]], w)
    _check(arr, "ewc::fisherB", fB, names, 7, 1e-5)
    _check(arr, "ewc::paramsB", pB, names, 7, 1e-6)
    _check(arr, "ewc::final_theta", dict(net.named_parameters()), names, 7, 1e-6)


def test_oracle_rw_flow_equals_reference(ref):
    """RW.py:128-265 executed by the reference == oracle.train.rw_* (running Fisher / scores, post-task normalisation)."""
    meta, arr = ref
    r = meta["rw_flow"]
    names, gnames = r["names"], r["stat_names"]
    net = OracleGenericUNet(1, 8, 3, 2)
    net.load_state_dict({n[6:]: torch.from_numpy(arr[n]) for n in arr.files if n.startswith("init::")})
    opt = otrain.make_optimizer(net)
    w = olosses.ds_loss_weights(2)
    st = otrain.rw_new_task_state(net)
    lA = []
    for b in ref_batches(3000, r["iters"]):
        lA.append(otrain.run_iteration(net, opt, b["data"], b["target"], w)[0])
        otrain.rw_update_f_s(net, st, r["alpha"], r["fisher_update_after"])
    assert np.allclose(lA, r["lossesA"], rtol=1e-6)
    fA, pA, sA = otrain.rw_finish_task(net, st, 1)
    _check(arr, "rw::fisherA", fA, gnames, 7, 1e-5)
    _check(arr, "rw::scoresA", sA, gnames, 7, 1e-5)
    _check(arr, "rw::paramsA", pA, names, 7, 1e-7)
    # RW.py:163-169: the dictionaries already hold the (zero) entry of the task being trained, which the loss omits (DS.py:106)
    fisher, params, scores = OrderedDict(taskA=fA, taskB=None), OrderedDict(taskA=pA, taskB=None), OrderedDict(taskA=sA, taskB=None)
    _fresh_head(net, arr)
    st = otrain.rw_new_task_state(net)
    # the reference hands named_parameters() over ONCE (RW.py:95-98): the penalty is live for the first forward only
    gen = [net.named_parameters()]
    lB = []
    for b in ref_batches(4000, r["iters"]):
        def pen():
            if gen:
                return olosses.rw_penalty(gen.pop(), fisher, params, scores, 0.4)
            return 0.0
        lB.append(otrain.run_iteration(net, opt, b["data"], b["target"], w, extra_loss=pen)[0])
        otrain.rw_update_f_s(net, st, r["alpha"], r["fisher_update_after"])
    assert np.allclose(lB, r["lossesB"], rtol=1e-6), (lB, r["lossesB"])
    fB, _, sB = otrain.rw_finish_task(net, st, 2)
    _check(arr, "rw::fisherB", fB, gnames, 7, 1e-5)
    _check(arr, "rw::scoresB", sB, gnames, 7, 1e-5)
    _check(arr, "rw::final_theta", dict(net.named_parameters()), names, 7, 1e-6)


class _Counting:
    def __init__(self, items):
        self.items, self.n = items, 0

    def __iter__(self):
        return self

    def __next__(self):
        b = self.items[self.n % len(self.items)]
        self.n += 1
        return b


def test_oracle_lwf_flow_equals_reference(ref):
    """HF.py:207-266 + LWF.py:298-370 executed by the reference == oracle.train.lwf_*: teacher logits of both heads, three
    phase-3 iterations (each consuming T + 2 = 4 batches of the generator), loss values (base + KL), updated weights."""
    meta, arr = ref
    f = meta["lwf_flow"]
    names = meta["ewc_flow"]["names"]
    net = OracleGenericUNet(1, 8, 3, 2)
    net.load_state_dict({n[6:]: torch.from_numpy(arr[n]) for n in arr.files if n.startswith("init::")})
    opt = otrain.make_optimizer(net)
    w = olosses.ds_loss_weights(2)
    lA = [otrain.run_iteration(net, opt, b["data"], b["target"], w)[0] for b in ref_batches(5000, 2)]
    assert np.allclose(lA, f["lossesA"], rtol=1e-6)
    head = lambda: OrderedDict((n, p.detach().clone()) for n, p in net.named_parameters() if n.startswith("seg_outputs."))
    heads = OrderedDict(taskA=head())
    _fresh_head(net, arr)                                  # add_new_task("taskB", use_init=True) + assemble_model
    heads["taskB"] = head()
    gT = _Counting(ref_batches(6000, 6))
    teach = otrain.lwf_target_logits(net, heads, gT, 3)
    assert gT.n == f["teacher_batches_consumed"] and list(teach.keys()) == f["teacher_tasks"]
    for t in teach:
        for i, lg in enumerate(teach[t]):
            exp = arr[f"lwf::teacher_{t}_{i}"]
            got = lg.numpy()[:, :, ::2, ::2, ::2]
            assert np.allclose(got, exp, rtol=1e-5, atol=1e-6), (t, i)
    gB = _Counting(ref_batches(7000, 12))
    lB = [otrain.lwf_iteration(net, opt, gB, heads, teach, i, w, f["T"]) for i in range(3)]
    assert gB.n == f["batches_consumed_B"] == 12
    assert np.allclose(lB, f["lossesB"], rtol=1e-5), (lB, f["lossesB"])
    _check(arr, "lwf::final_theta", dict(net.named_parameters()), names, 7, 1e-6)


def test_oracle_forward_wiring_matches_the_reference_forward_lines(golden_dir):
    """tests/golden/forward_wiring_reference.npz: ``Generic_ViT_UNet.forward`` of the reference (generic_ViT_UNet.py:216-284 -- upstream's
    U-Net forward "copied from original implementation" around the transformer) called unbound on the oracle network with the
    transformer stood in by the identity (oracle/make_goldens_forward.py).  The oracle's own ``forward`` must return the same tensors
    bit for bit: skip indexing, ``cat((x, skip), 1)`` order, decoder level -> segmentation layer, order of the deep-supervision tuple,
    and the single-output mode."""
    d = np.load(golden_dir + "/forward_wiring_reference.npz")
    net = OracleGenericUNet(*[int(v) for v in d["ctor"]])
    net.load_state_dict({k[4:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("sd::")})
    net.eval()
    x = torch.from_numpy(d["x"])
    with torch.no_grad():
        out = net(x)
        net.do_ds = False
        single = net(x)
    n_levels = int(d["ctor"][3])
    assert isinstance(out, tuple) and len(out) == n_levels
    for i in range(n_levels):
        ref = torch.from_numpy(d[f"ref_ds_{i}"])
        assert out[i].shape == ref.shape and float((out[i] - ref).abs().max()) <= 1e-6 * float(ref.abs().max()), i
    assert float((single - torch.from_numpy(d["ref_single"])).abs().max()) <= 1e-6 * float(single.abs().max())
    assert [tuple(o.shape[2:]) for o in out] == [(8, 16, 8), (4, 8, 4), (2, 4, 2)]      # full resolution first
