"""fp16 parameter shadow for the fp16-autocast steps of the zoo: ONE multi-tensor cast per step instead of one per parameter.

Under `torch.autocast` every `F.conv*` call casts its fp32 weight and bias to fp16 (a 5 us launch each) and autograd casts the
fp16 gradients back (another launch each).  The SSND2Net step has ~850 plain torch convolutions (GSC gates, patch embeddings,
side / fuse convolutions): ~3 400 cast launches, 16 ms of a 218 ms step (profiles/r03_ssnd2net_graph_kernels.txt: 6 677 + 5 146
`float16_copy` / `float16tofloat32_copy` launches in three steps).  The reference pays the same casts
(/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainer.py:1128-1139 wraps the forward in autocast).

`ParamShadow(network)(x)` casts the eligible parameters with ONE `torch._foreach_copy_` (a handful of multi-tensor launches) inside
an autograd Function - its backward casts all fp16 gradients back the same way - and runs the network with the fp16 tensors
substituted through `torch.func.functional_call`; autocast then finds fp16 operands and casts nothing.  Numerics are unchanged
(the same fp32 -> fp16 rounding of the same values, the same fp16 -> fp32 widening of the gradients).
Eligible: parameters owned directly by plain torch convolution modules (incl. `common2d._Conv2d`), outside the REBNCONV / RSU4F
sub-trees of common2d.py AND u2net.py (those run on the HIP conv path, which packs its weights from the fp32 master itself).  Everything else - TokenLinear,
LayerNorm, SS2D / SSND parameters - is read as fp32 by hand-written kernels and is left alone."""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
from torch import nn


class _ShadowCast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *params):
        outs = [torch.empty_like(p, dtype=torch.float16) for p in params]
        torch._foreach_copy_(outs, [p.detach() for p in params])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        idx = [i for i, g in enumerate(grads) if g is not None]
        wide = [torch.empty_like(grads[i], dtype=torch.float32) for i in idx]
        if idx:
            torch._foreach_copy_(wide, [grads[i] for i in idx])
        out = [None] * len(grads)
        for i, w in zip(idx, wide):
            out[i] = w
        return tuple(out)


def _eligible(network: nn.Module) -> Tuple[List[str], List[nn.Parameter]]:
    from .nets.common2d import REBNCONV, RSU4F, _Conv2d
    from .nets.u2net import REBNCONV as _U2REBNCONV
    kinds = (nn.Conv1d, nn.Conv2d, nn.Conv3d, nn.ConvTranspose2d, nn.ConvTranspose3d, _Conv2d)
    # every class whose convolution runs on the HIP conv path (nnuzoo_amd/rebnconv.py packs the fp32 master weight itself)
    hip_conv = (REBNCONV, RSU4F, _U2REBNCONV)
    skip = {id(m) for root in network.modules() if isinstance(root, hip_conv) for m in root.modules()}
    # the depthwise convolution of an SS2D block is not a torch call either: csrc/ss2d_dwconv.hip reads its fp32 weight (an fp16
    # shadow was cast straight back to fp32 there: 248 launches per SSND2Net step)
    from .nets.m2net import SS2D
    skip |= {id(root.conv2d) for root in network.modules() if isinstance(root, SS2D) and hasattr(root, "conv2d")}
    from .nets.ssnd import SSND
    skip |= {id(root.convnd.conv) for root in network.modules()
             if isinstance(root, SSND) and root.spatial_dims == 2 and hasattr(root.convnd, "conv")}
    from .nets.u2net_multi import Convolution as _MonaiUnit
    skip |= {id(root.conv) for root in network.modules() if isinstance(root, _MonaiUnit) and root.hip_capable()}
    # convolutions a net marked as read by hand-written kernels from the fp32 master (1x1 patch embeddings / stage outputs on the token
    # Linear kernels, side / fuse heads: nets/m2net.py)
    skip |= {id(m) for m in network.modules() if getattr(m, "_nnz_fp32_master", False)}
    names, params = [], []
    for mname, m in network.named_modules():
        if id(m) in skip or type(m) not in kinds:
            continue
        for pname, p in m.named_parameters(recurse=False):
            if p.dtype == torch.float32 and p.requires_grad and p.is_cuda:
                names.append(f"{mname}.{pname}" if mname else pname)
                params.append(p)
    return names, params


class ParamShadow:
    def __init__(self, network: nn.Module):
        self.network = network
        self.enabled = os.environ.get("NNZ_PARAM_SHADOW", "1") != "0"
        self.last_count = 0

    def __call__(self, x: torch.Tensor):
        if not (self.enabled and x.is_cuda and torch.is_autocast_enabled()
                and torch.get_autocast_dtype("cuda") == torch.float16 and torch.is_grad_enabled()):
            return self.network(x)
        names, params = _eligible(self.network)
        self.last_count = len(params)
        if not params:
            return self.network(x)
        shadows = _ShadowCast.apply(*params)
        return torch.func.functional_call(self.network, dict(zip(names, shadows)), (x,))
