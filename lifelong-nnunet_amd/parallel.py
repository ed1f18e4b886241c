"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce of the flat gradient arena over xGMI.

The reference has no distributed code at all; the oracle for this layer is "N ranks on
shards == 1 rank on the concatenated batch" (patches are independent units: InstanceNorm is per sample,
CE is a voxel mean, per-sample Dice is a sample mean -- SURVEY.md 8e).

Design for xGMI (point-to-point links, 7 x ~153 GB/s per GPU): few large messages.  The arena is laid out
in forward order, backward completes it back-to-front, so buckets are cut from the TAIL and each is
all-reduced on a side HIP stream as soon as the backward plan passes its lower bound (an event orders
the side stream after the producing kernels).  The 1/world averaging is folded into the optimiser's
unscale factor -- no extra pass over the gradients.

Two choices are unmeasured on xGMI (no N > 1 hardware was available to the builder) and are therefore switches, so that the
driver's scaling run can be repeated with the alternative without a code change:
  LNN_DP_BUCKET_MB  bucket size in MB (default 32: 4 buckets for the 124.8 MB arena of the 160x192x160 plan)
  LNN_DP_STREAM     ``wgrad`` (default): a bucket is all-reduced FROM the engine's weight-gradient stream -- measured on ONE GPU with
                    a world-1 RCCL group (+0.04 ms); on N ranks an RCCL kernel that waits for a peer then sits in front of the next
                    weight gradient on that stream.  ``own``: the exchange has its own stream (a third stream sharing the chip).
``stats()`` says what happened in the last step (buckets launched from inside backward / by ``finish``) and, with
``collect_timing``, how long the main stream actually waited for the exchange (``exposed_comm_ms``) -- bench.py reports both.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def make_buckets(size: int, bucket_elems: int) -> List[Tuple[int, int]]:
    """Tail-first partition of [0, size) into ranges of <= bucket_elems elements."""
    out, hi = [], size
    while hi > 0:
        lo = max(0, hi - bucket_elems)
        out.append((lo, hi))
        hi = lo
    return out


def _env_bucket_bytes(default: int) -> int:
    v = os.environ.get("LNN_DP_BUCKET_MB")
    if not v:
        return default
    mb = float(v)
    if not mb > 0:
        raise ValueError(f"LNN_DP_BUCKET_MB={v!r}: a positive number of megabytes is expected")
    return int(mb * (1 << 20))


def _env_stream_mode() -> str:
    v = os.environ.get("LNN_DP_STREAM", "wgrad")
    if v not in ("wgrad", "own"):
        raise ValueError(f"LNN_DP_STREAM={v!r}: 'wgrad' or 'own'")
    return v


class GradAllReducer:
    def __init__(self, grad: torch.Tensor, process_group=None, bucket_bytes: int = None, overlap: bool = True,
                 force: bool = False, stream_mode: str = None):
        self.grad = grad
        bucket_bytes = _env_bucket_bytes(32 << 20) if bucket_bytes is None else bucket_bytes
        self.bucket_bytes = bucket_bytes
        self.stream_mode = _env_stream_mode() if stream_mode is None else stream_mode
        assert self.stream_mode in ("wgrad", "own")
        self.collect_timing = False      # bench.py: HIP events around finish()'s wait (two records per step, no synchronisation)
        self._timing = []                # [(event before the wait, event after it)] of the last steps
        self._in_backward = self._in_finish = 0
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = dist.is_initialized() and (self.world > 1 or force)   # force: exercise the path on one rank
        self.buckets = make_buckets(grad.numel(), max(1, bucket_bytes // grad.element_size()))
        # host tensors (gloo, CPU tests): the same watermark-driven bucket order, each all-reduce simply runs in line
        self.overlap = overlap
        self._stream = None     # created on first use: the buckets normally leave from the engine's weight-gradient stream
        self.next = 0
        self._used = set()

    @property
    def stream(self):
        if self._stream is None and self.grad.is_cuda:
            self._stream = torch.cuda.Stream()
        return self._stream

    @property
    def averaging_factor(self) -> float:
        return 1.0 / self.world

    def begin(self):
        self.next = 0
        self._used = set()
        self._in_backward = self._in_finish = 0

    def _launch(self, lo, hi, stream=None):
        if not self.active:
            return
        view = self.grad[lo:hi]
        stream = (stream if stream is not None else self.stream) if self.grad.is_cuda else None
        if stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            stream.wait_event(ev)
            with torch.cuda.stream(stream):
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg)
            self._used.add(stream)
        else:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg)

    def progress(self, watermark: int, stream=None):
        """All gradient elements at offsets >= watermark are final once the work enqueued so far on the CURRENT stream
        and on ``stream`` has run.  ``stream``: the engine's weight-gradient side stream -- the bucket is all-reduced
        FROM that stream (after an event of the current one), so the exchange queues behind the weight gradients it
        needs and no third stream competes with the two that already share the chip."""
        if not self.overlap:
            return
        producer = stream
        if self.stream_mode == "own":
            stream = None                # _launch then uses this object's own stream, ordered behind the current one ...
        while self.next < len(self.buckets) and self.buckets[self.next][0] >= watermark:
            if stream is None and producer is not None and self.grad.is_cuda and self.active:
                self.stream.wait_stream(producer)      # ... and behind the weight gradients of the bucket on the engine's side stream
            self._launch(*self.buckets[self.next], stream=stream)
            self.next += 1
            self._in_backward += 1

    def finish(self):
        while self.next < len(self.buckets):
            self._launch(*self.buckets[self.next])
            self.next += 1
            self._in_finish += 1
        if self.active:
            timed = self.collect_timing and self.grad.is_cuda and len(self._used) > 0
            if timed:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            for st in self._used:
                torch.cuda.current_stream().wait_stream(st)
            if timed:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                self._timing.append((e0, e1))
                del self._timing[:-64]

    def stats(self, synchronize: bool = False):
        """What the last step did: buckets launched from inside backward (``progress``) and by ``finish``; with ``collect_timing``
        the mean time the step's stream waited for the exchange in ``finish`` over the recorded steps (needs the events to have
        completed: ``synchronize=True`` waits for the device)."""
        out = {"buckets": len(self.buckets), "bucket_mb": self.bucket_bytes / (1 << 20), "stream": self.stream_mode,
               "buckets_sent_in_backward": self._in_backward, "buckets_sent_by_finish": self._in_finish}
        if self._timing:
            if synchronize:
                torch.cuda.synchronize()
            ms = [a.elapsed_time(b) for a, b in self._timing if b.query()]
            if ms:
                out["exposed_comm_ms"] = sum(ms) / len(ms)
                out["exposed_comm_ms_max"] = max(ms)
                out["exposed_comm_steps"] = len(ms)
        return out


def all_reduce_stats(t: torch.Tensor, process_group=None):
    """Small-message helper (batch-Dice TP/FP/FN, Fisher arenas in true-accumulate mode)."""
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=process_group)
    return t
