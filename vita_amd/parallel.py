"""One process per GPU.  Two things shard on this path:

  * ODE inference: rows are independent (no op on the path mixes rows; LayerNorm is per row), so rank r takes the
    contiguous rows shard_range(B, r, N) and there is NO data-path collective.
  * Training: data parallel.  Each rank's vita_compute_loss_f32 writes the gradient of ITS shard's mean loss into a
    flat fp32 buffer laid out [decoder | flow net: rest, then layer by layer | encoders] and records an event per
    segment (VitaPolicy.reduce_plan gives the order they fire in); BucketReducer all-reduces a segment on a side stream
    as soon as its event fires, so the reductions overlap the rest of the backward.  The sum is turned into the mean
    (x 1/N) inside the fused clip+AdamW+EMA kernel, not by a separate pass.

The reducer works on any torch.distributed backend: "nccl" (= RCCL over xGMI on the GPU box) and "gloo" on CPU
tensors, which is how the tests cover world_size 2 without GPUs.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced row shard [lo, hi) of rank `rank`: the first n_rows % world ranks get one extra row."""
    if world < 1 or not (0 <= rank < world) or n_rows < 0:
        raise ValueError(f"bad shard request n_rows={n_rows} rank={rank} world={world}")
    q, r = divmod(n_rows, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def world_info(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


class BucketReducer:
    """Sum-all-reduce contiguous segments of one flat buffer, each as soon as it is ready."""

    def __init__(self, flat: torch.Tensor, ranges: Sequence[Tuple[int, int]], group=None, max_bucket_elems: Optional[int] = None,
                 rehearse: bool = False):
        if flat.dim() != 1 or not flat.is_contiguous():
            raise ValueError("BucketReducer needs a flat contiguous buffer")
        self.flat, self.group = flat, group
        self.rank, self.world = world_info(group)
        # rehearse: issue the collectives even in a one-rank group (a sum over one rank is the identity) so the RCCL calls,
        # the side stream and the event waits can be exercised on a one-GPU box
        self.active = self.world > 1 or (rehearse and dist.is_available() and dist.is_initialized())
        # split oversized segments so a single RCCL call stays below max_bucket_elems (xGMI ring collectives are
        # per-link bound; a few tens of MB per call keeps the pipeline full without serialising on one giant op)
        self.segments: List[List[Tuple[int, int]]] = []
        for (lo, hi) in ranges:
            if not (0 <= lo <= hi <= flat.numel()):
                raise ValueError(f"segment ({lo}, {hi}) outside the buffer of {flat.numel()} elements")
            parts, step = [], (max_bucket_elems or max(hi - lo, 1))
            for s in range(lo, hi, step):
                parts.append((s, min(hi, s + step)))
            self.segments.append(parts)
        self.cuda = flat.is_cuda
        self.stream = torch.cuda.Stream(device=flat.device) if (self.cuda and self.active) else None
        self._handles = []
        # gloo has no device-side collectives: a GPU buffer is staged through pinned host memory by THIS class (explicit
        # copies on the side stream + gloo's CPU all-reduce) rather than by ProcessGroupGloo's own CUDA path, which hung
        # intermittently with two ranks on one device.  RCCL ("nccl") reduces in place on the device.
        self.host_staged = bool(self.cuda and self.active and dist.get_backend(group) == "gloo")
        self._pinned = None
        if self.host_staged:
            n = max((hi - lo for parts in self.segments for (lo, hi) in parts), default=0)
            self._pinned = torch.empty(n, dtype=flat.dtype).pin_memory()

    def reduce(self, i: int, ready_event: Optional["torch.cuda.Event"] = None) -> None:
        """Start the all-reduce of segment i (after `ready_event` on the side stream when given)."""
        if not self.active:
            return
        if self.cuda:
            if ready_event is not None:
                self.stream.wait_event(ready_event)
            else:
                self.stream.wait_stream(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self.stream):
                for (lo, hi) in self.segments[i]:
                    if hi <= lo:
                        continue
                    if self.host_staged:
                        host = self._pinned[:hi - lo]
                        host.copy_(self.flat[lo:hi], non_blocking=True)
                        self.stream.synchronize()
                        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                        self.flat[lo:hi].copy_(host, non_blocking=True)
                        self.stream.synchronize()     # the pinned buffer is reused by the next piece
                    else:
                        dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
        else:
            for (lo, hi) in self.segments[i]:
                if hi > lo:
                    self._handles.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self) -> float:
        """Make the compute stream wait for every started reduction; returns the factor (1/world) that turns the sums
        into means."""
        if self.active:
            if self.cuda:
                torch.cuda.current_stream(self.flat.device).wait_stream(self.stream)
            else:
                for h in self._handles:
                    h.wait()
                self._handles = []
        return 1.0 / self.world
