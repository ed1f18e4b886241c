"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce of the flat gradient arena over xGMI.

The reference has no distributed code at all; the oracle for this layer is "N ranks on
shards == 1 rank on the concatenated batch" (patches are independent units: InstanceNorm is per sample,
CE is a voxel mean, per-sample Dice is a sample mean -- SURVEY.md 8e).

Design for xGMI (point-to-point links, 7 x ~153 GB/s per GPU): few large messages.  The arena is laid out
in forward order, backward completes it back-to-front, so buckets are cut from the TAIL and each is
all-reduced on a side HIP stream as soon as the backward plan passes its lower bound (an event orders
the side stream after the producing kernels).  The 1/world averaging is folded into the optimiser's
unscale factor -- no extra pass over the gradients.
"""
from __future__ import annotations

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


class GradAllReducer:
    def __init__(self, grad: torch.Tensor, process_group=None, bucket_bytes: int = 32 << 20, overlap: bool = True,
                 force: bool = False):
        self.grad = grad
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
        while self.next < len(self.buckets) and self.buckets[self.next][0] >= watermark:
            self._launch(*self.buckets[self.next], stream=stream)
            self.next += 1

    def finish(self):
        while self.next < len(self.buckets):
            self._launch(*self.buckets[self.next])
            self.next += 1
        if self.active:
            for st in self._used:
                torch.cuda.current_stream().wait_stream(st)


def all_reduce_stats(t: torch.Tensor, process_group=None):
    """Small-message helper (batch-Dice TP/FP/FN, Fisher arenas in true-accumulate mode)."""
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=process_group)
    return t
