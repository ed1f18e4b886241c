"""-m gpu: the data-parallel step on ONE GPU (``LNN_FORCE_DP=1``: a world-size-1 RCCL group, every collective really
issued).  A 2-rank run of the HIP engine is not possible on the 1-GPU test box, so the single-process run pins what can
be pinned there: the watermark order (buckets final tail-first, each launched exactly once, from the weight-gradient side
stream), equality with the plain (non-DP) step, the batch-Dice exchange and the accumulated-Fisher mode through the
forced-DP code paths.  The 2-rank arithmetic of GradAllReducer itself runs on CPU/gloo in tests/test_parallel_gloo.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

from lifelong_nnunet_amd import get_trainer_class                          # noqa: E402
from lifelong_nnunet_amd.synthetic import make_patch_batch                 # noqa: E402

DEV = "cuda:0"
TOY = {"patch_size": (16, 32, 16), "batch_size": 2, "num_pool": 2, "base_num_features": 8, "num_classes": 3,
       "num_input_channels": 1, "synthetic_period": 4}


@pytest.fixture(scope="module")
def nccl_world1():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    yield
    dist.destroy_process_group()


def _trainer(ext, force_dp, **kw):
    os.environ["LNN_FORCE_DP"] = "1" if force_dp else "0"
    try:
        # ordered weight-gradient reductions: without them two fp16 runs differ by the order of the fp32 atomics, which a few
        # fp16 rounding flips amplify to 1e-3 in the smallest Fisher entries -- with them plain and forced-DP steps must
        # agree BIT FOR BIT (a race between the streams would show)
        tr = get_trainer_class(ext)("seg_outputs", "taskA", plans=dict(TOY), device=DEV, deterministic_wgrad=True, **kw)
        tr.initialize(True, num_epochs=1)
    finally:
        os.environ["LNN_FORCE_DP"] = "0"
    tr.num_batches_per_epoch, tr.num_val_batches_per_epoch = 3, 0
    return tr


def _batches(n, seed=50):
    out = []
    for i in range(n):
        d, t = make_patch_batch(2, TOY["patch_size"], 2, seed=seed + i)
        out.append({"data": d, "target": t, "keys": ["a", "b"]})
    return out


def test_forced_dp_step_equals_plain_step(nccl_world1):
    plain, dp = _trainer("sequential", False), _trainer("sequential", True)
    assert plain.dp is None and dp.dp is not None and dp.dp.active and dp.dp.world == 1
    dp.network.load_state_dict(plain.network.state_dict())
    dp.mh_network.update_after_iteration()
    # small buckets so that several become final DURING backward
    from lifelong_nnunet_amd.parallel import make_buckets
    dp.dp.buckets = make_buckets(dp.network.arena.grad.numel(), 4096)
    log = []
    orig_progress, orig_launch = dp.dp.progress, dp.dp._launch

    def progress(wm, stream=None):
        log.append(("wm", wm, stream))
        return orig_progress(wm, stream)

    def launch(lo, hi, stream=None):
        log.append(("ar", lo, hi, stream))
        return orig_launch(lo, hi, stream)
    dp.dp.progress, dp.dp._launch = progress, launch
    bs = _batches(2)
    for b in bs:
        lp = plain.run_iteration(iter([b]), True)
        ld = dp.run_iteration(iter([b]), True)
        assert float(lp) == float(ld)
    tp, td = plain.network.arena.theta, dp.network.arena.theta
    assert torch.equal(tp, td)                                 # ordered reductions everywhere: bit-identical
    wms = [e[1] for e in log if e[0] == "wm"]
    ars = [(e[1], e[2]) for e in log if e[0] == "ar"]
    per_step = len(dp.dp.buckets)
    assert len(ars) == 2 * per_step
    first = ars[:per_step]
    assert first == dp.dp.buckets                               # tail-first, each bucket exactly once, full cover
    assert first[0][1] == dp.network.arena.grad.numel() and first[-1][0] == 0
    assert all(a >= b for a, b in zip(wms[:len(wms) // 2], wms[1:len(wms) // 2]))        # watermarks never move up within a backward
    # buckets that became final during backward were launched from the engine's weight-gradient side stream
    eng = list(dp.network._engines.values())[0]
    during = [e for e in log if e[0] == "ar" and e[3] is not None]
    assert during and all(e[3] is eng._sides[0] for e in during)


def test_batch_dice_and_accumulated_fisher_through_forced_dp(nccl_world1):
    """batch_dice=True: the tp/fp/fn exchange inside the loss (world 1: identity) must leave the loss equal to the
    non-DP one; fisher_mode='accumulate': mean of the squared per-batch gradients, all-reduced once per task."""
    a = _trainer("ewc", False, batch_dice=True, fisher_mode="accumulate")
    b = _trainer("ewc", True, batch_dice=True, fisher_mode="accumulate")
    b.network.load_state_dict(a.network.state_dict())
    b.mh_network.update_after_iteration()
    data = _batches(6, seed=70)
    for tr in (a, b):
        tr.data_provider = lambda task, split, plans: iter(data)
        tr.reinitialize("taskA")
        tr.run_training("taskA")
    assert a.all_tr_losses == b.all_tr_losses
    names = list(a.fisher["taskA"].keys())
    fa = torch.cat([a.fisher["taskA"][n].reshape(-1) for n in names if a.fisher["taskA"][n].numel() > 1])
    fb = torch.cat([b.fisher["taskA"][n].reshape(-1) for n in names if b.fisher["taskA"][n].numel() > 1])
    assert float(fa.sum()) > 0 and torch.equal(fa, fb)
