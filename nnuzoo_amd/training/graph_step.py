"""hipGraph capture of the forward + loss + backward part of a training step.

The zoo models issue 7 000 - 10 000 small kernels per step (profiles/r01_m2net_step_kernels_v6.txt): eager launch
overhead, not GPU time, sets the step time.  MI355X-first answer (task brief: "capture launch-bound inner loops in hipGraphs"):
record the step once on a side stream and replay it as ONE graph launch.  torch.cuda.CUDAGraph is hipGraph on ROCm;
our C-ABI launchers are capture-safe by construction (stream-ordered, allocation-free, no host sync, and no
hipMemsetAsync: a captured memset node replays with a corrupted fill pattern on this stack once the process has made
further allocations - every zeroing is a kernel, csrc/common.hpp zero_async; tests/test_graph_replay_gpu.py).
Library ops captured alongside (ATen's multi-block reductions zero their semaphores with hipMemsetAsync, some MIOpen /
hipBLASLt paths too) are handled by a graph REWRITING pass: the captured hipGraph_t is kept (`keep_graph=True`), every
memset node is replaced by a fill-kernel node with the same edges (csrc/graph_tools.hip: nnz_graph_replace_memsets), and
only then the graph is instantiated - the replayed graph consists of kernel nodes only.

What is captured: zero-grad-free forward, loss, (scaled) backward into static .grad buffers.  What stays eager: the
GradScaler unscale / inf check, clip_grad_norm_, optimizer step and the loss read-back - the reference's train_step
semantics (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainer.py:1128-1144) are unchanged.
"""
from __future__ import annotations

import itertools
import os
from typing import Callable, List, Optional

import torch

_CAPTURE_TOKENS = itertools.count(1)      # one token per capture, process-wide (grads_token)


def capture_memset_free(fn: Callable, stream: torch.cuda.Stream):
    """Captures fn() on `stream` into a hipGraph, rewrites its memset nodes into kernel nodes and instantiates it.
    Returns (torch.cuda.CUDAGraph, number of memset nodes rewritten).  Capture on the SAME side stream the warm-up ran
    on: library handles (MIOpen / rocBLAS keep one per stream) are then already initialised; a fresh capture stream made
    their lazy set-up run inside the capture (segfault)."""
    import ctypes as C
    from .._lib import call
    if os.environ.get("NNZ_GRAPH_KEEP", "1") == "0":      # diagnostics: plain capture, no rewriting
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            fn()
        return graph, 0
    graph = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(graph, stream=stream):
        fn()
    raw = graph.raw_cuda_graph()
    n = C.c_int(0)
    if os.environ.get("NNZ_GRAPH_REWRITE", "1") != "0":   # diagnostics switch
        call("nnz_graph_replace_memsets", C.c_void_p(int(raw)), C.byref(n))
    graph.instantiate()
    return graph, int(n.value)


class GraphedForwardBackward:
    def __init__(self, network: torch.nn.Module, loss_fn: Callable, grad_scaler, autocast: bool, warmup_iters: int = 2,
                 forward_fn: Optional[Callable] = None):
        self.network, self.loss_fn, self.scaler, self.autocast = network, loss_fn, grad_scaler, autocast
        self.forward_fn = forward_fn if forward_fn is not None else network    # e.g. nnuzoo_amd.param_shadow.ParamShadow
        self.warmup_iters = warmup_iters
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.static_data = None
        self.static_target: List[torch.Tensor] = []
        self.static_loss = None
        self.memset_nodes_replaced = 0
        self._key = None
        self._static_grads = []        # [(parameter, the .grad tensor the captured backward writes)]
        self._static_arena = None      # explicit-schedule networks: (arena, layout, unused ids) the captured backward fills
        self.generation = 0            # number of captures so far: (id(self), generation) names one set of static gradients
        self._token = None             # None = "no valid capture": FusedAdamW rescans the gradient addresses

    def _loss(self, out, target):
        if isinstance(out, (tuple, list)):
            return self.loss_fn(list(out), target)
        return self.loss_fn(out, target[0] if isinstance(target, (tuple, list)) else target)   # single-output network

    def _fwd_bwd(self, data, target):
        if self.autocast:
            with torch.autocast('cuda'):
                loss = self._loss(self.forward_fn(data), target)
        else:
            loss = self._loss(self.forward_fn(data), target)
        from ..token_linear import deferred_wgrads
        with deferred_wgrads():      # fp32 Linear weight gradients of the pass run as ONE grouped launch at the end
            (self.scaler.scale(loss) if self.scaler is not None else loss).backward()
        return loss

    def _capture(self, data, target):
        self._token = None             # a failed re-capture must not leave the previous capture's token standing
        self.static_data = data.clone()
        self.static_target = [t.clone() for t in target]
        params = [p for p in self.network.parameters()]
        # the eager warm-up passes must leave no trace in the module state: BatchNorm running estimates / counters are
        # restored afterwards (the captured pass itself is only recorded, not executed)
        buffers = [(b, b.detach().clone()) for b in self.network.buffers()]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup_iters):  # eager warm-up: MIOpen/rocBLAS find, lazy attribute set-up, allocator
                for p in params:
                    p.grad = None
                self._fwd_bwd(self.static_data, self.static_target)
        torch.cuda.current_stream().wait_stream(side)
        for p in params:
            p.grad = None
        self.graph, self.memset_nodes_replaced = capture_memset_free(
            lambda: setattr(self, "static_loss", self._fwd_bwd(self.static_data, self.static_target)), side)
        with torch.no_grad():
            for b, saved in buffers:
                b.copy_(saved)
        # the captured backward WRITES (p.grad was None at capture: assignment, not accumulation) into these tensors on
        # every replay; whatever happens to p.grad between replays (optimizer.zero_grad(set_to_none=True), an eager step
        # in between, user hooks), __call__ re-attaches them so that clip / optimizer never see None or a stale buffer
        self._static_grads = [(p, p.grad) for p in params if p.grad is not None]
        # Networks with an explicit backward schedule (PlainConvUNet) publish their gradients through a flat arena that the
        # Python side of `_run_backward` registers on the module (`_last_arena` & co).  A replay runs no Python, so after an
        # EAGER step in between (use_hip_graph toggled, a validation pass with gradients, ...) the module would still point
        # at that eager pass's arena while the replay fills the captured one - FusedSGD would then apply stale gradients.
        # The captured arena is therefore remembered here and re-registered after every replay.
        if hasattr(self.network, "grad_arena") and self.network.grad_arena() is not None:
            self._static_arena = (self.network._last_arena, self.network._arena_layout, self.network._last_unused)
        self._key = (tuple(data.shape), tuple(tuple(t.shape) for t in target))
        self.generation += 1
        self._token = next(_CAPTURE_TOKENS)

    def grads_token(self):
        """identifies the static gradient tensors the replay writes (see FusedAdamW.fused_step).  Drawn from a process-wide
        counter at every capture: `id(self)` can be handed to a NEW instance after this one is freed (the encoder-freezing hook
        and the re-initialisation path drop and rebuild the captured step), and a recycled (id, generation) pair would let the
        optimizer keep a chunk table that points into the destroyed graph's pool."""
        return self._token

    def __call__(self, data: torch.Tensor, target: List[torch.Tensor]) -> torch.Tensor:
        """Runs forward+backward for (data, target); gradients are in p.grad afterwards.  Returns the loss tensor."""
        key = (tuple(data.shape), tuple(tuple(t.shape) for t in target))
        if self.graph is None or key != self._key:
            self._capture(data, target)
        self.static_data.copy_(data, non_blocking=True)
        for s, t in zip(self.static_target, target):
            s.copy_(t, non_blocking=True)
        self.graph.replay()
        for p, g in self._static_grads:
            if p.grad is not g:
                p.grad = g
        if self._static_arena is not None:
            net = self.network
            net._last_arena, net._arena_layout, net._last_unused = self._static_arena
        return self.static_loss


class GraphedDDPStep:
    """The data-parallel step of an explicit-schedule network (PlainConvUNet) as hipGraph SEGMENTS with the RCCL collectives
    between them (round 4).

    Under DDP the step used to run eagerly, because the bucketed all-reduce is launched from inside the backward schedule so
    that it overlaps the remaining backward - at ~300 launches per step that costs 3.6 % against the replayed N = 1 step before
    any communication.  Capturing the collectives INSIDE one graph would remove that too, but a captured RCCL launch cannot be
    tested on this one-GPU pool, and a hang at N = 8 is not an acceptable failure mode.  So the collectives stay ordinary
    stream-ordered launches and the graph is cut where the schedule hands a bucket over:

        capture   forward + loss + backward run once on a side stream; every time the reducer has a full bucket
                  (`BucketedAllReduce.capture_boundary`) the current capture ends and the next one begins (same memory pool) ->
                  segments S_0 .. S_k and the arena slices [lo_i, hi_i) that are final after S_i
        replay    for i: S_i.replay(); all_reduce(arena[lo_i:hi_i], async)      (the collective waits for S_i on RCCL's stream and
                  runs beside S_i+1)      then wait for all; the SUM is turned into the mean inside the fused optimizer
                  (inv_scale / world): no extra pass over the arena.

    The network part runs WITHOUT autograd: `_run_forward` -> leaf tensors for the logits -> loss + its backward (autograd, loss
    ops only) -> `_run_backward(rec, dlogits)` in this thread, so that the segment boundaries are ordinary Python calls on the
    capturing thread.  Parameter `.grad`s are not populated; the step must be finished by `FusedSGD.fused_step`, which reads
    the gradient arena (the trainer checks that).  Requires a loss without collectives (batch_dice False: 3d_fullres)."""

    def __init__(self, network: torch.nn.Module, loss_fn: Callable, grad_scaler, warmup_iters: int = 2):
        if getattr(network, "grad_reducer", None) is None or not hasattr(network, "_run_backward"):
            raise ValueError("GraphedDDPStep needs an explicit-schedule network with an attached BucketedAllReduce")
        self.network, self.loss_fn, self.scaler, self.warmup_iters = network, loss_fn, grad_scaler, warmup_iters
        self.segments: List[torch.cuda.CUDAGraph] = []
        self.slices: List[Optional[tuple]] = []       # slice to reduce after segment i (None: nothing)
        self.static_data = None
        self.static_target: List[torch.Tensor] = []
        self.static_loss = None
        self._static_arena = None
        self._key = None
        self.memset_nodes_replaced = 0

    def _fwd_bwd(self, data, target):
        net = self.network
        with torch.no_grad():
            outs, rec = net._run_forward(data.float().contiguous(), save=True)
        leaves = [o.detach().requires_grad_(True) for o in outs]
        loss = self.loss_fn(leaves if net.decoder.deep_supervision else leaves[0], target)
        (self.scaler.scale(loss) if self.scaler is not None else loss).backward()
        with torch.no_grad():
            net._run_backward(rec, [l.grad for l in leaves])
        return loss.detach()

    def _capture(self, data, target):
        import ctypes as C
        from .._lib import call
        net, red = self.network, self.network.grad_reducer
        self.static_data = data.clone()
        self.static_target = [t.clone() for t in target]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup_iters):          # eager warm-up WITH the real collectives (every rank takes part)
                self._fwd_bwd(self.static_data, self.static_target)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        pool = torch.cuda.graph_pool_handle()
        self.segments, self.slices = [], []
        cur = {"g": None}

        def begin():
            g = torch.cuda.CUDAGraph(keep_graph=True)
            g.capture_begin(pool=pool, capture_error_mode="relaxed")   # the loss backward runs on autograd's thread
            cur["g"] = g

        def end(sl):
            g = cur["g"]
            g.capture_end()
            n = C.c_int(0)
            call("nnz_graph_replace_memsets", C.c_void_p(int(g.raw_cuda_graph())), C.byref(n))
            self.memset_nodes_replaced += int(n.value)
            g.instantiate()
            self.segments.append(g)
            self.slices.append(sl)

        def boundary(lo, hi, final):
            end((lo, hi) if hi > lo else None)
            if not final:
                begin()

        red.capture_boundary = boundary
        done = False
        try:
            with torch.cuda.stream(side):
                begin()
                self.static_loss = self._fwd_bwd(self.static_data, self.static_target)
            done = True
        finally:
            red.capture_boundary = None
            if not done:
                # an exception between begin() and the final boundary: the open capture is ended (its graph dropped)
                # and nothing half-captured survives - the next call captures from scratch
                g = cur["g"]
                try:
                    with torch.cuda.stream(side):
                        if g is not None and torch.cuda.is_current_stream_capturing():
                            g.capture_end()
                except Exception:        # the original exception is the one to surface
                    pass
                self.segments, self.slices, self._key = [], [], None
        torch.cuda.current_stream().wait_stream(side)
        self._static_arena = (net._last_arena, net._arena_layout, net._last_unused)
        self._key = (tuple(data.shape), tuple(tuple(t.shape) for t in target))

    def __call__(self, data: torch.Tensor, target: List[torch.Tensor]) -> torch.Tensor:
        key = (tuple(data.shape), tuple(tuple(t.shape) for t in target))
        if not self.segments or key != self._key:
            self._capture(data, target)
        self.static_data.copy_(data, non_blocking=True)
        for s, t in zip(self.static_target, target):
            s.copy_(t, non_blocking=True)
        net, red = self.network, self.network.grad_reducer
        arena = self._static_arena[0]
        handles = []
        for g, sl in zip(self.segments, self.slices):
            g.replay()
            if sl is not None:
                handles.append(red.reduce_slice_async(arena, sl[0], sl[1]))
        for h in handles:
            h.wait()
        red.slices_last_step = [sl for sl in self.slices if sl is not None]
        red.buckets_last_step = len(red.slices_last_step)
        red.bytes_last_step = sum(h - l for l, h in red.slices_last_step) * arena.element_size()
        net._last_arena, net._arena_layout, net._last_unused = self._static_arena
        return self.static_loss
