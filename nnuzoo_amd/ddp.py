"""Data-parallel gradient exchange for the explicit-schedule networks: one process per GPU, RCCL over xGMI.

The reference wraps the network in torch DDP (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainer.py:278-280)
and relies on autograd hooks to overlap the bucketed NCCL all-reduce with backward.  Our backward is ONE autograd
node with an explicit reverse schedule, so the overlap is explicit too: after every decoder/encoder stage the
schedule hands the finished parameter gradients to `BucketedAllReduce.stage_done`, which launches an asynchronous
all-reduce (backend "nccl" == RCCL on ROCm; its own stream) as soon as a bucket is full, while the next stage's
kernels keep the compute stream busy.  `finish` waits (stream-ordered, no host sync on RCCL) and averages.
The PlainConvUNet schedule writes every parameter gradient into ONE flat arena in completion order, so a bucket is
a slice of that arena reduced IN PLACE (`stage_done_arena` / `finish_arena`): no flatten / unflatten copies (the
dict form below costs two extra passes over the 125 MB of gradients and is kept for generic callers).

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of S bytes costs ~2*(7/8)*S/153 GB/s,
i.e. ~1.4 ms for the 124.8 MB of fp32 gradients of the 3d_fullres PlainConvUNet - buckets of >= 16 MB keep the
per-collective latency (~20-30 us) negligible while still giving 6-8 collectives to overlap with backward.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.distributed as dist


class BucketedAllReduce:
    def __init__(self, params: List[torch.nn.Parameter], process_group=None, bucket_bytes: int = 16 << 20):
        self.group = process_group
        self.bucket_bytes = bucket_bytes
        self.world = dist.get_world_size(process_group)
        self._seen = set()
        self._pending: List[torch.Tensor] = []
        self._pending_bytes = 0
        self._inflight = []
        self._arena_lo = 0
        self._slices = []              # (lo, hi) element ranges of the arena collectives launched in the current step
        self.slices_last_step = []     # ... of the most recent finished step (bench.py / tests read these)
        self.buckets_last_step = 0
        self.bytes_last_step = 0       # bytes all-reduced by the most recent finished step (arena form)
        # hipGraph capture of the data-parallel step (training/graph_step.py GraphedDDPStep): while set, a bucket that is
        # ready is REPORTED through this callable - boundary(lo, hi, final) - instead of being all-reduced: the capture ends
        # the current graph segment there and the replay launches the collective for the slice between two segments
        self.capture_boundary = None

    def _launch(self):
        if not self._pending:
            return
        flat = torch.cat([g.reshape(-1) for g in self._pending])
        handle = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._inflight.append((handle, flat, self._pending))
        self._pending, self._pending_bytes = [], 0

    def stage_done(self, grads: Dict[torch.nn.Parameter, torch.Tensor]):
        """Called by the backward schedule with the dict of all gradients computed so far."""
        for p, g in grads.items():
            if p in self._seen:
                continue
            self._seen.add(p)
            self._pending.append(g)
            self._pending_bytes += g.numel() * g.element_size()
        if self._pending_bytes >= self.bucket_bytes:
            self._launch()

    # ---- arena form: gradients live in one flat buffer in completion order; buckets are slices of it --------------
    def stage_done_arena(self, arena: torch.Tensor, filled: int):
        """Everything in arena[:filled] is final.  Launch an in-place all-reduce for the not-yet-reduced part once it
        is at least one bucket long."""
        lo = self._arena_lo
        if (filled - lo) * arena.element_size() >= self.bucket_bytes:
            if self.capture_boundary is not None:
                self.capture_boundary(lo, filled, False)
                self._slices.append((lo, filled))
                self._arena_lo = filled
                return
            h = dist.all_reduce(arena[lo:filled], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._inflight.append((h, None, None))
            self._slices.append((lo, filled))
            self._arena_lo = filled

    def finish_arena(self, arena: torch.Tensor, filled: int):
        lo = self._arena_lo
        if self.capture_boundary is not None:
            if filled > lo:
                self._slices.append((lo, filled))
            self.capture_boundary(lo, filled, True)
            self._arena_lo = 0
            self.slices_last_step, self._slices = self._slices, []
            self.buckets_last_step = len(self.slices_last_step)
            self.bytes_last_step = sum(h - l for l, h in self.slices_last_step) * arena.element_size()
            return
        if filled > lo:
            h = dist.all_reduce(arena[lo:filled], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._inflight.append((h, None, None))
            self._slices.append((lo, filled))
        for h, _, _ in self._inflight:
            h.wait()
        arena[:filled].mul_(1.0 / self.world)
        self._inflight = []
        self._arena_lo = 0
        self.slices_last_step, self._slices = self._slices, []
        self.buckets_last_step = len(self.slices_last_step)
        self.bytes_last_step = sum(h - l for l, h in self.slices_last_step) * arena.element_size()

    def reduce_slice_async(self, arena: torch.Tensor, lo: int, hi: int):
        """replay side of a captured step: the in-place SUM all-reduce of one recorded bucket (no averaging - the fused
        optimizer folds 1 / world into its unscale factor)"""
        return dist.all_reduce(arena[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self, all_grads: List[torch.Tensor]):
        """Flush, wait for every collective and write the averaged gradients back in place."""
        seen_ids = {id(g) for _, _, gs in self._inflight for g in gs} | {id(g) for g in self._pending}
        for g in all_grads:  # gradients that never went through stage_done (e.g. zero-filled unused heads)
            if id(g) not in seen_ids:
                self._pending.append(g)
        self._launch()
        inv = 1.0 / self.world
        for handle, flat, gs in self._inflight:
            handle.wait()
            off = 0
            for g in gs:
                n = g.numel()
                g.copy_(flat[off:off + n].view_as(g))
                g.mul_(inv)
                off += n
        self._inflight = []
        self._seen = set()


def attach_bucketed_allreduce(network, process_group=None, bucket_bytes: int = 12 << 20) -> BucketedAllReduce:
    """DDP for nnuzoo_amd networks: broadcast rank-0 parameters, then reduce gradients inside backward."""
    params = list(network.parameters())
    with torch.no_grad():
        for p in params:
            dist.broadcast(p.data, src=0, group=process_group)
        for b in network.buffers():
            dist.broadcast(b.data, src=0, group=process_group)
    red = BucketedAllReduce(params, process_group, bucket_bytes)
    network.grad_reducer = red
    return red


def allreduce_gradients(params, process_group=None, bucket_bytes: int = 64 << 20) -> int:
    """Gradient averaging for the autograd-graph networks (the X^2-Net zoo), called once after backward: the existing
    gradients are flattened into buckets, all-reduced (SUM) asynchronously, and written back divided by the world size.
    Parameters without a gradient are skipped: every rank runs the same graph, so the same parameters are without one
    everywhere (the zoo's inner `seg_layers` never receive gradients - the reference needs torch DDP's
    find_unused_parameters for these nets).  Their steps are launch-bound (7-10 thousand small kernels), so the exchange
    (160 MB for M2Net = ~1.5 ms over xGMI) is not overlapped with backward.  Returns the number of gradients reduced."""
    from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors
    world = dist.get_world_size(process_group)
    grads = [p.grad for p in params if p.grad is not None]
    if world == 1 or not grads:
        return len(grads)
    buckets, cur, cur_bytes = [], [], 0
    by_type = {}
    for g in grads:                                  # one dtype / device per flat buffer
        by_type.setdefault((g.dtype, g.device), []).append(g)
    for gs in by_type.values():
        for g in gs:
            cur.append(g)
            cur_bytes += g.numel() * g.element_size()
            if cur_bytes >= bucket_bytes:
                buckets.append(cur)
                cur, cur_bytes = [], 0
        if cur:
            buckets.append(cur)
            cur, cur_bytes = [], 0
    inflight = []
    for gs in buckets:
        flat = _flatten_dense_tensors(gs)
        inflight.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=process_group, async_op=True), flat, gs))
    inv = 1.0 / world
    for handle, flat, gs in inflight:
        handle.wait()
        flat.mul_(inv)
        torch._foreach_copy_(gs, list(_unflatten_dense_tensors(flat, gs)))
    return len(grads)


def prepare_autograd_network_for_ddp(network: torch.nn.Module, process_group=None) -> torch.nn.Module:
    """What the reference's initialize() does before wrapping in torch DDP (nnUNetTrainer.py:275-280): BatchNorm ->
    SyncBatchNorm (RSU4F / REBNCONV stages of the zoo), and every rank starts from rank 0's parameters and buffers."""
    network = torch.nn.SyncBatchNorm.convert_sync_batchnorm(network, process_group)
    with torch.no_grad():
        for t in list(network.parameters()) + list(network.buffers()):
            dist.broadcast(t.data, src=0, group=process_group)
    return network
