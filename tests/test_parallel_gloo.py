"""world_size-2 gloo tests (CPU): the bucketed, watermark-driven gradient all-reduce and the data-parallel
equivalence it relies on -- N ranks on shards == 1 rank on the concatenated batch."""
import os
import socket

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lifelong_nnunet_amd.parallel import GradAllReducer, make_buckets


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_buckets_cover_arena_tail_first():
    b = make_buckets(1000, 300)
    assert b == [(700, 1000), (400, 700), (100, 400), (0, 100)]
    assert make_buckets(10, 100) == [(0, 10)]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import losses
    from oracle.unet import OracleGenericUNet
    from lifelong_nnunet_amd.synthetic import make_patch_batch
    torch.manual_seed(5)
    net = OracleGenericUNet(1, 8, 3, 2)
    w = losses.ds_loss_weights(2)
    data, tgts = make_patch_batch(world, (8, 16, 8), 2, seed=3)
    # --- local shard: one patch per rank
    d, t = data[rank:rank + 1], [x[rank:rank + 1] for x in tgts]
    losses.multiple_output_loss(net(d), t, w).backward()
    params = [p for p in net.parameters() if p.grad is not None]
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    red = GradAllReducer(flat, bucket_bytes=4096 * 4)
    assert len(red.buckets) > 3
    red.begin()
    # simulate backward progress: watermarks walk from the tail to the head of the arena
    launched = []
    orig = red._launch
    red._launch = lambda lo, hi, stream=None: (launched.append((lo, hi)), orig(lo, hi, stream))[1]
    seen_before_finish = 0
    for wm in range(flat.numel(), -1, -flat.numel() // 7):
        red.progress(wm)
        assert all(lo >= wm for lo, hi in launched)          # only buckets that lie entirely behind the watermark
        seen_before_finish = len(launched)
    assert 0 < seen_before_finish <= len(red.buckets)
    red.progress(0)                                          # backward has reached the head of the arena
    assert len(launched) == len(red.buckets)                 # the watermark walk itself drove the whole exchange (not finish())
    red.finish()
    assert launched == red.buckets                           # tail-first, each bucket exactly once
    flat *= red.averaging_factor
    if rank == 0:
        net.zero_grad()
        losses.multiple_output_loss(net(data), tgts, w).backward()
        full = torch.cat([p.grad.reshape(-1) for p in params])
        out["rel"] = float((flat - full).norm() / full.norm())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_full_batch_gradient():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out["rel"] < 1e-5      # fp32 round-off only


def test_data_parallel_switches_are_read_from_the_environment(monkeypatch):
    """LNN_DP_BUCKET_MB / LNN_DP_STREAM (parallel.py): defaults = the shipped plan (32 MB buckets, all-reduce from the weight-gradient
    stream); bad values are errors, not silently ignored."""
    import torch
    from lifelong_nnunet_amd.parallel import GradAllReducer
    g = torch.zeros(3 * (1 << 20))                         # 12 MB of fp32
    monkeypatch.delenv("LNN_DP_BUCKET_MB", raising=False)
    monkeypatch.delenv("LNN_DP_STREAM", raising=False)
    r = GradAllReducer(g)
    assert r.stream_mode == "wgrad" and r.bucket_bytes == 32 << 20 and len(r.buckets) == 1
    monkeypatch.setenv("LNN_DP_BUCKET_MB", "4")
    monkeypatch.setenv("LNN_DP_STREAM", "own")
    r = GradAllReducer(g)
    assert r.stream_mode == "own" and len(r.buckets) == 3 and r.buckets[0] == (2 << 20, 3 << 20)
    assert r.stats() == {"buckets": 3, "bucket_mb": 4.0, "stream": "own", "buckets_sent_in_backward": 0, "buckets_sent_by_finish": 0}
    r.begin(); r.progress(2 << 20); r.finish()
    assert r.stats()["buckets_sent_in_backward"] == 1 and r.stats()["buckets_sent_by_finish"] == 2
    monkeypatch.setenv("LNN_DP_STREAM", "third")
    with pytest.raises(ValueError):
        GradAllReducer(g)
    monkeypatch.setenv("LNN_DP_STREAM", "wgrad")
    monkeypatch.setenv("LNN_DP_BUCKET_MB", "-1")
    with pytest.raises(ValueError):
        GradAllReducer(g)
