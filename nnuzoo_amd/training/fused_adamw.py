"""AdamW whose step - together with GradScaler.unscale_ and clip_grad_norm_ - runs as two HIP launches over a device table.

`FusedAdamW` IS a `torch.optim.AdamW` (param_groups for the LR scheduler; `state[p]` holds `step`, `exp_avg`, `exp_avg_sq` under
torch's names, so state_dict / load_state_dict and the reference's checkpoints - nnUNetTrainer.py:1291-1352 - keep working),
constructed with the hyper-parameters of the reference's X^2-Net plugins (AdamW(lr 1e-4, weight_decay 5e-2, eps 1e-5),
/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainerM2Net.py:58-65).  `fused_step(inv_scale, max_norm)` replaces

    grad_scaler.unscale_(optimizer); clip_grad_norm_(params, 12); grad_scaler.step(optimizer)

of train_step (nnUNetTrainer.py:1133-1137) by csrc/optimizer.hip `nnz_adamw_fused`: with 1 500 - 4 100 parameter tensors torch's
multi-tensor kernels are ~260 launches per step, each behind 60-100 us of host-side list handling (8-17 ms of GPU idle per step
in the round-3 traces).  The moments are views into two flat tensors; the per-parameter step counters are the elements of one
device vector.  The chunk table holds raw addresses of parameters, gradients and moments: they are static while the step is
replayed as a hipGraph (training/graph_step.py re-attaches the captured gradient tensors); when an address moved (eager steps
allocate new gradients, load_state_dict, .to()) the table is rebuilt - vectorised with numpy, ~2 ms for 4 000 tensors.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from .. import _lib
from .._lib import call, ptr, stream_ptr
from ..hip_ops import det_scratch

CHUNK = 16384  # elements per workgroup of the update kernel


class FusedAdamW(torch.optim.AdamW):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        # fused=True keeps torch's own fallback path on its device-side multi-tensor kernel (step counters on the device)
        params = list(params)
        use_torch_fused = all(p.is_cuda for p in params)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=use_torch_fused)
        self._table: Optional[torch.Tensor] = None
        self._nchunks = 0
        self._ptrs = None            # (param ptrs, grad ptrs, m ptrs, v ptrs) of the table in use
        self._flat_m: Optional[torch.Tensor] = None
        self._flat_v: Optional[torch.Tensor] = None
        self._steps: Optional[torch.Tensor] = None
        self._members: List[torch.nn.Parameter] = []
        self.table_builds = 0        # diagnostics / tests
        self._token = None

    # ---- availability --------------------------------------------------------------------------------------------------------
    def fused_available(self) -> bool:
        """one parameter group with plain AdamW options on the device (dtype / layout of every tensor is checked when the chunk
        table is built - that raises, it does not fall back silently)"""
        g = self.param_groups
        if len(g) != 1 or g[0].get("amsgrad") or g[0].get("maximize") or g[0].get("differentiable"):
            return False
        p0 = g[0]["params"][0]
        return p0.is_cuda

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._token = None
        self._flat_m = None           # restored moments are fresh tensors: re-linked into the flat buffers on the next step

    # ---- state as flat buffers ----------------------------------------------------------------------------------------------
    def _build_state(self, members: List[torch.nn.Parameter]):
        dev = members[0].device
        total = sum(p.numel() for p in members)
        flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        steps = torch.zeros(len(members), dtype=torch.float32, device=dev)
        off = 0
        for i, p in enumerate(members):
            st = self.state[p]
            n = p.numel()
            m, v = flat_m[off:off + n].view_as(p), flat_v[off:off + n].view_as(p)
            if "exp_avg" in st:          # moments restored from a checkpoint or produced by earlier torch steps
                m.copy_(st["exp_avg"])
                v.copy_(st["exp_avg_sq"])
                steps[i] = float(st["step"])
            st["exp_avg"], st["exp_avg_sq"], st["step"] = m, v, steps[i]
            off += n
        self._flat_m, self._flat_v, self._steps, self._members = flat_m, flat_v, steps, list(members)

    def _state_linked(self, members) -> bool:
        """cheap per-step check (identity of the member list); load_state_dict invalidates explicitly"""
        if self._flat_m is None or len(members) != len(self._members):
            return False
        return all(p is q for p, q in zip(members, self._members))

    def _build_table(self, members, pp, gp):
        nb = int(_lib.load().nnz_adam_chunk_bytes())
        assert nb == 40
        for p in members:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.dtype == torch.float32
                    and p.grad.is_contiguous()):
                raise _lib.HipCallError("FusedAdamW: parameters and gradients must be contiguous float32 device tensors")
        sizes = np.array([p.numel() for p in members], dtype=np.int64)
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
        nck = (sizes + CHUNK - 1) // CHUNK
        owner = np.repeat(np.arange(len(members)), nck)                      # chunk -> parameter
        first = np.concatenate([[0], np.cumsum(nck)[:-1]])
        within = (np.arange(int(nck.sum())) - np.repeat(first, nck)) * CHUNK      # element offset inside the parameter
        n = np.minimum(CHUNK, sizes[owner] - within).astype(np.int32)
        rec = np.zeros(len(owner), dtype=np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i4"),
                                                   ("owner", "<i4")]))
        boff = (4 * within).astype(np.uint64)            # (uint64 + int64 would promote to float64)
        rec["p"] = pp[owner] + boff
        rec["g"] = gp[owner] + boff
        rec["m"] = self._flat_m.data_ptr() + 4 * (offs[owner] + within)
        rec["v"] = self._flat_v.data_ptr() + 4 * (offs[owner] + within)
        rec["n"] = n
        rec["owner"] = owner.astype(np.int32)          # the kernel takes the bias corrections from steps[owner]
        self._table = torch.from_numpy(rec.view(np.uint8).copy()).to(members[0].device)
        self._nchunks = len(owner)
        self._ptrs = (pp.copy(), gp.copy())
        self.table_builds += 1

    # ---- the fused tail -------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def fused_step(self, inv_scale: Optional[torch.Tensor], max_norm: float, grads_token=None) -> torch.Tensor:
        """unscale + clip + AdamW on the parameters' current .grad tensors.  inv_scale: 1-element fp32 device tensor (1 / loss
        scale) or None.  Returns the 1-element fp32 `found_inf` tensor (> 0: the step was skipped).

        grads_token: a caller's promise that parameters and gradients are the SAME tensors as at the previous call with an
        equal token (graph replay: training/graph_step.py hands out (id, capture generation)); the per-step address scan of
        every parameter - ~4 ms of host time for the 4 100 tensors of SSND2Net - is then skipped."""
        g = self.param_groups[0]
        if grads_token is not None and grads_token == self._token and self._table is not None and self._flat_m is not None:
            members = self._members
        else:
            members = [p for p in g["params"] if p.grad is not None]
            if not members:
                raise _lib.HipCallError("FusedAdamW.fused_step: no parameter has a gradient")
            if not self._state_linked(members):
                self._build_state(members)
                self._ptrs = None
            pp = np.fromiter((p.data_ptr() for p in members), dtype=np.uint64, count=len(members))
            gp = np.fromiter((p.grad.data_ptr() for p in members), dtype=np.uint64, count=len(members))
            if self._ptrs is None or not (np.array_equal(pp, self._ptrs[0]) and np.array_equal(gp, self._ptrs[1])):
                self._build_table(members, pp, gp)
            self._token = grads_token
        dev = members[0].device
        stats = torch.empty(2, dtype=torch.float32, device=dev)
        sc = det_scratch(dev, 3)
        b1, b2 = g["betas"]
        call("nnz_adamw_fused", ptr(self._table), self._nchunks, ptr(stats), ptr(sc.acc), ptr(sc.counter), ptr(inv_scale),
             float(max_norm), float(g["lr"]), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]),
             ptr(self._steps), int(self._steps.numel()), stream_ptr())
        return stats[1:2]
