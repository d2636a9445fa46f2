"""SGD(momentum, nesterov, weight decay) whose step runs as the fused HIP tail over the gradient arena.

`FusedSGD` IS a `torch.optim.SGD` (param_groups for the LR scheduler, `state[p]['momentum_buffer']`, state_dict /
load_state_dict for the reference's checkpoints: nnUNetTrainer.py:1291-1352), constructed like the reference's optimizer
(nnUNetTrainer.configure_optimizers, nnUNetTrainer.py:501-506).  When the network exposes the flat fp32 gradient arena
its backward schedule fills (`grad_arena()`, nnuzoo_amd/nets/plain_conv_unet.py), `fused_step` replaces

    grad_scaler.unscale_(optimizer); clip_grad_norm_(params, 12); grad_scaler.step(optimizer)

of train_step (nnUNetTrainer.py:1133-1137) by two kernels (csrc/optimizer.hip).  Momentum buffers are views into one
flat tensor in arena order.  Networks without an arena keep using the inherited torch step.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import torch

from .. import _lib
from .._lib import call, ptr, stream_ptr
from ..hip_ops import det_scratch

CHUNK = 16384  # elements per workgroup of the update kernel


class FusedSGD(torch.optim.SGD):
    def __init__(self, network: torch.nn.Module, lr: float, weight_decay: float = 0.0, momentum: float = 0.99,
                 nesterov: bool = True):
        super().__init__(network.parameters(), lr, weight_decay=weight_decay, momentum=momentum, nesterov=nesterov)
        if not nesterov or momentum <= 0:
            raise ValueError("FusedSGD implements the reference's configuration: momentum > 0 with nesterov=True")
        self._net = network
        self._flat_mom: Optional[torch.Tensor] = None
        self._chunks: Optional[torch.Tensor] = None
        self._nchunks = 0
        self._links: List[Tuple[torch.nn.Parameter, int, int, int]] = []  # (param, offset, param ptr, momentum ptr)
        self._total = 0
        self._unused = frozenset()

    # ---- availability --------------------------------------------------------------------------------------------
    def fused_available(self) -> bool:
        return getattr(self._net, "grad_arena", None) is not None and self._net.grad_arena() is not None

    def fused_available_static(self) -> bool:
        """the network publishes a gradient arena at all (before the first backward has produced one)"""
        return getattr(self._net, "grad_arena", None) is not None

    # ---- flat momentum + chunk table (rebuilt when storage moved: first step, load_state_dict, .to()) ------------
    def _build(self, layout: List[Tuple[torch.nn.Parameter, int]], device, unused=None):
        lib = _lib.load()
        if unused is None:
            unused = self._net_unused()
        self._total = sum(p.numel() for p, _ in layout)
        flat = torch.zeros(self._total, dtype=torch.float32, device=device)
        # parameters without a gradient in this schedule (unused deep-supervision heads): no momentum state, no update -
        # what torch.optim.SGD does for a parameter whose .grad is None
        layout = [(p, off) for p, off in layout if id(p) not in unused]
        for p, off in layout:
            st = self.state[p]
            old = st.get('momentum_buffer')
            view = flat[off:off + p.numel()].view_as(p)
            if old is not None:
                view.copy_(old)  # buffers restored from a checkpoint or produced by earlier torch steps
            st['momentum_buffer'] = view
        nb = lib.nnz_sgd_chunk_bytes()
        recs = []
        for p, off in layout:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.HipCallError("FusedSGD: parameters must be contiguous float32 tensors")
            n = p.numel()
            for s in range(0, n, CHUNK):
                recs.append((p.data_ptr() + 4 * s, flat.data_ptr() + 4 * (off + s), off + s, min(CHUNK, n - s)))
        host = (C.c_char * (nb * len(recs)))()
        for i, (pp, mp, aoff, n) in enumerate(recs):
            _lib.check(lib.nnz_sgd_chunk_fill(C.byref(host, i * nb), C.c_void_p(pp), C.c_void_p(mp), aoff, n),
                       "nnz_sgd_chunk_fill")
        self._chunks = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(device)
        self._nchunks = len(recs)
        self._flat_mom = flat
        self._links = [(p, off, p.data_ptr(), self.state[p]['momentum_buffer'].data_ptr()) for p, off in layout]
        self._unused = unused

    def _linked(self) -> bool:
        if self._flat_mom is None or self._unused != self._net_unused():
            return False
        for p, _, pptr, mptr in self._links:
            mb = self.state[p].get('momentum_buffer')
            if mb is None or mb.data_ptr() != mptr or p.data_ptr() != pptr:
                return False
        return True

    def _net_unused(self):
        f = getattr(self._net, "grad_arena_unused", None)
        return f() if f is not None else frozenset()

    # ---- the fused tail ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def fused_step(self, inv_scale: Optional[torch.Tensor], max_norm: float) -> torch.Tensor:
        """Runs unscale + clip + SGD on the network's gradient arena.  inv_scale: 1-element fp32 device tensor
        (1 / loss scale) or None.  Returns the 1-element fp32 `found_inf` tensor (> 0: the step was skipped)."""
        arena = self._net.grad_arena()
        if arena is None:
            raise _lib.HipCallError("FusedSGD.fused_step: the network has no gradient arena (run backward first)")
        if not self._linked():
            self._build(self._net.grad_arena_layout(), arena.device, self._net_unused())
        g = self.param_groups[0]
        stats = torch.empty(2, dtype=torch.float32, device=arena.device)
        sc = det_scratch(arena.device, 2)      # fixed-point cross-workgroup sum: the clip factor is reproducible run to run
        call("nnz_grad_sumsq_nonfinite_det", ptr(arena), self._total, ptr(stats), ptr(sc.acc), ptr(sc.counter), stream_ptr())
        call("nnz_sgd_nesterov_fused", ptr(self._chunks), self._nchunks, ptr(arena), ptr(stats), ptr(inv_scale),
             float(max_norm), float(g['lr']), float(g['momentum']), float(g['weight_decay']), 0, stream_ptr())
        return stats[1:2]

    def total_grad_norm(self, inv_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
        """unscaled global gradient norm of the current arena (diagnostics / tests)"""
        arena = self._net.grad_arena()
        stats = torch.empty(2, dtype=torch.float32, device=arena.device)
        sc = det_scratch(arena.device, 2)
        call("nnz_grad_sumsq_nonfinite_det", ptr(arena), arena.numel(), ptr(stats), ptr(sc.acc), ptr(sc.counter),
             stream_ptr())
        n = stats[0].sqrt()
        return n * inv_scale[0] if inv_scale is not None else n
