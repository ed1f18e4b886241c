"""-m gpu: ``python bench.py --gpus N`` launches its own ranks (VERDICT r3 "next" 2).  Two ranks on the one leased GPU over gloo
(``--share-gpu``: a plumbing check, never a measurement): the JSON line must report what the collective library saw."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_gpus_2_launches_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu",
                        "--steps", "2", "--warmup", "1", "--workload", "c1", "--no-cpu-baseline", "--no-roofline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2
    assert d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    assert d["backend"] == "gloo" and d["distinct_devices"] == 1
    assert d["value"] > 0 and d["steps"] == 2
    # VERDICT r5 task 7: the N > 1 line says what the gradient exchange did
    dp = d["data_parallel"]
    assert dp["stream"] == "wgrad" and dp["bucket_mb"] == 32 and dp["buckets"] >= 1
    assert dp["buckets_sent_in_backward"] == dp["buckets"] and dp["buckets_sent_by_finish"] == 0
    assert dp["exposed_comm_ms"] is not None and len(dp["exposed_comm_ms_per_rank"]) == 2
    assert d["ms_per_step_rank_min"] <= d["ms_per_step_rank_max"] <= d["ms_per_step"] * 1.0001 + 1e-6


@pytest.mark.gpu
def test_bench_data_parallel_switches_select_the_alternatives():
    """LNN_DP_BUCKET_MB / LNN_DP_STREAM (parallel.py): smaller buckets on the exchange's own stream -- same JSON contract, more buckets,
    every one of them still launched from inside backward."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LNN_DP_BUCKET_MB="4", LNN_DP_STREAM="own")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu",
                        "--steps", "2", "--warmup", "1", "--workload", "c1", "--no-cpu-baseline", "--no-roofline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    dp = d["data_parallel"]
    assert dp["stream"] == "own" and dp["bucket_mb"] == 4 and dp["buckets"] == 6          # 22.4 MB arena of the 40x56x40 plan
    assert dp["buckets_sent_in_backward"] == 6 and dp["buckets_sent_by_finish"] == 0 and d["ranks_seen"] == 2


def test_bench_refuses_a_world_that_is_not_gpus():
    """A launcher that exports WORLD_SIZE=1 while asking for 8 GPUs must not record a 1-GPU number as n_gpus 8 (CPU test)."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in r.stderr and "--gpus 8" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
