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


def test_bench_refuses_a_world_that_is_not_gpus():
    """A launcher that exports WORLD_SIZE=1 while asking for 8 GPUs must not record a 1-GPU number as n_gpus 8 (CPU test)."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in r.stderr and "--gpus 8" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
