"""world_size-2 gloo test (CPU) of the TRAINERS' data-parallel wiring: two processes run ``run_iteration`` (plain, EWC with a live
penalty, RW) and EWC's ``after_train`` with the launch layer replaced by recorders (tests/test_plan_dryrun.py: no kernel runs), but with
a REAL gloo all-reduce behind ``parallel.GradAllReducer``.  Each rank's gradient arena is filled with ``rank + 1`` when its backward
starts; after the iteration every element must be 1 + 2 = 3 on both ranks -- every bucket exchanged exactly once, from inside backward
(``finish()`` finds nothing left), the 1 / world factor handed to the optimiser, and no exchange where the trainer says there is none
(EWC's Fisher in accumulate mode)."""
import os
import socket
from collections import OrderedDict

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


PLANS = {"patch_size": (16, 16, 16), "batch_size": 2, "num_pool": 2, "base_num_features": 8, "num_classes": 3, "num_input_channels": 1}


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from lifelong_nnunet_amd import get_trainer_class
    from lifelong_nnunet_amd import native as nat
    from lifelong_nnunet_amd.parallel import make_buckets
    from tests.test_plan_dryrun import install_dry_run
    rec = install_dry_run(setattr)
    res = {}
    for ext in ("sequential", "ewc", "rw"):
        tr = get_trainer_class(ext)("seg_outputs", "A", plans=dict(PLANS), device="cpu")
        tr.initialize(True, num_epochs=1)
        assert tr.dp is not None and tr.dp.active and tr.dp.world == world
        arena = tr.network.arena
        tr.dp.buckets = make_buckets(arena.size, 4096)       # the toy arena in more than one bucket
        assert len(tr.dp.buckets) > 3
        if ext == "ewc":                # a previous task: the penalty node is in the loss graph, its gradient lands in the arena first
            tr.fisher["prev"] = OrderedDict((n, torch.ones_like(p)) for n, p in tr.network.named_parameters())
            tr.params["prev"] = OrderedDict((n, p.detach().clone()) for n, p in tr.network.named_parameters())
            tr.loss.update_ewc_params(tr.fisher, tr.params)
            assert len(tr.loss.tasks) == 1
        state = {"fill": False, "launched": [], "in_finish": False}
        begin, finish, launch = tr.dp.begin, tr.dp.finish, tr.dp._launch

        def begin_and_arm(state=state, begin=begin):
            begin()
            state["fill"] = True
            state["launched"] = []

        def counted_launch(lo, hi, stream=None, state=state, launch=launch):
            state["launched"].append((lo, hi, state["in_finish"]))
            return launch(lo, hi, stream)

        def watched_finish(state=state, finish=finish):
            state["in_finish"] = True
            finish()
            state["in_finish"] = False

        def fake_call(name, *args, state=state, arena=arena):
            rec.append(("call", name, "?", args))
            if state["fill"]:           # the first launch after dp.begin(): this rank's gradient of the step
                arena.grad.fill_(float(rank + 1))
                state["fill"] = False
        nat.call = fake_call
        tr.dp.begin, tr.dp.finish, tr.dp._launch = begin_and_arm, watched_finish, counted_launch
        for it in range(2):
            tr.run_iteration(tr.tr_gen, True)
            assert bool((arena.grad == 3.0).all()), (ext, it, arena.grad.unique())
            assert [b[:2] for b in state["launched"]] == tr.dp.buckets and not any(b[2] for b in state["launched"]), (ext, it)
            assert tr.network.on_grad_progress is None
            assert abs(tr.last_inv_scale * tr.amp_grad_scaler.get_scale() - 0.5) < 1e-12
        res[ext] = len(state["launched"])
        # VERDICT r5 task 7: the exchange reports what it did, in both stream modes (LNN_DP_STREAM; on host tensors the mode only
        # changes which stream object is handed on -- the bucket order and counts must not depend on it)
        st = tr.dp.stats()
        assert st["stream"] == "wgrad" and st["buckets_sent_in_backward"] == len(tr.dp.buckets) and st["buckets_sent_by_finish"] == 0
        tr.dp.stream_mode = "own"
        tr.run_iteration(tr.tr_gen, True)
        assert bool((arena.grad == 3.0).all()), (ext, "own")
        st = tr.dp.stats()
        assert st["stream"] == "own" and st["buckets_sent_in_backward"] == len(tr.dp.buckets) and st["buckets_sent_by_finish"] == 0
        assert [b[:2] for b in state["launched"]] == tr.dp.buckets
        tr.dp.stream_mode = "wgrad"
        if ext == "ewc":
            tr.num_batches_per_epoch = 3
            for mode in ("parity", "accumulate"):
                tr.fisher_mode = mode
                tr.fisher[tr.task], tr.params[tr.task] = OrderedDict(), OrderedDict()
                state["launched"] = []
                rec.clear()
                if mode == "accumulate":        # no exchange: nothing arms the fill, the arena keeps this rank's own values
                    arena.grad.fill_(float(rank + 1))
                tr.after_train()
                assert tr.network.on_grad_progress is None
                if mode == "parity":
                    # (after_train zeroes the gradients at its end: the exchanged values are seen through the square's launch)
                    assert [b[:2] for b in state["launched"]] == tr.dp.buckets and not any(b[2] for b in state["launched"])
                    sq = [r for r in rec if r[0] == "call" and r[1] == "lnn_fisher_square"]
                    unscale = 1.0 if tr.fisher_keeps_loss_scale else 1.0 / tr.amp_grad_scaler.get_scale()
                    assert len(sq) == 1 and abs(sq[0][3][3] - 0.5 * unscale) < 1e-12       # 1 / world folded into the square
                else:
                    assert state["launched"] == []
                    assert len([r for r in rec if r[0] == "call" and r[1] == "lnn_fisher_accumulate"]) == 3
                assert set(tr.fisher[tr.task]) == {n for n, _ in tr.network.named_parameters()}
            res["ewc_after_train"] = True
    if rank == 0:
        out.update(res)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_trainers_exchange_every_bucket_once_from_inside_backward():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out["sequential"] > 3 and out["ewc"] == out["sequential"] == out["rw"] and out["ewc_after_train"]
