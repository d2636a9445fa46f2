"""All stochastic-depth draws of one forward pass from ONE launch.

The reference's DropPath draws a (B, 1, 1, 1) tensor per block and branch (nets/swt2net.py:395-409 `torch.rand`, timm's
`bernoulli_` in nets/m2net.py): ~240 launches of ~4.7 us per SwT2Net step, ~80 per M2Net step
(profiles/r05_swt2net_graph_kernels.txt: 554 `distribution_elementwise_grid_stride_kernel` launches in 2.7 steps) for 2 floats each.
Here the outer network opens a `DrawTable` around its forward; the first pass counts the requests (and serves them one by one, as
before), every later pass draws `torch.rand(requests, B)` once and hands out rows.  A 0 / 1 mask is floor(keep + u) of a uniform u -
the reference's own formula in swt2net.py, and a Bernoulli(keep) variable like timm's `bernoulli_(keep)` - formed inside the
residual kernels (csrc/residual.hip, dense32 epilogues) from the row.  The VALUES differ from the per-block call sequence (one
Philox call instead of many), the distribution and the per-sample independence do not; CPU tensors and networks without a table
keep the per-block calls.  Inside a captured hipGraph the single `torch.rand` is a graph-safe generator call like the ones it replaces.
NNZ_DROPPATH_TABLE=0 restores the per-block draws."""
from __future__ import annotations

import os

import torch

ENABLED = os.environ.get("NNZ_DROPPATH_TABLE", "1") != "0"
_ACTIVE = []          # stack of open tables (nested networks: the outermost one serves)


class DrawTable:
    def __init__(self, owner: torch.nn.Module, batch: int, device: torch.device):
        self.owner, self.B, self.device = owner, int(batch), device
        self.rows = None
        self.used = 0

    def __enter__(self):
        if ENABLED and not _ACTIVE and self.device.type == "cuda" and self.owner.training:
            want = int(self.owner.__dict__.get("_droppath_requests", 0))
            if want > 0:
                self.rows = torch.rand((want, self.B), dtype=torch.float32, device=self.device)
        _ACTIVE.append(self)
        return self

    def __exit__(self, *exc):
        _ACTIVE.pop()
        if not _ACTIVE and self.owner.training:
            self.owner.__dict__["_droppath_requests"] = self.used
        return False


def uniform(batch: int, device: torch.device) -> torch.Tensor:
    """B fp32 uniform [0, 1) draws, contiguous: a row of the open table or a `torch.rand` call of its own"""
    if _ACTIVE:
        t = _ACTIVE[0]
        t.used += 1
        if t.rows is not None and t.used <= t.rows.shape[0] and batch == t.B and device == t.rows.device:
            return t.rows[t.used - 1]
    return torch.rand((batch,), dtype=torch.float32, device=device)
