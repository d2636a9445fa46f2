"""PlainConvUNet ("nnUNet") for MI355X: same constructor, attributes and state_dict layout as
dynamic_network_architectures.architectures.unet.PlainConvUNet (0.3.x), which the reference instantiates by name
in /root/reference/nnunetv2/utilities/get_network_from_plans.py:18-62 from the planner's arch kwargs
(/root/reference/nnunetv2/experiment_planning/experiment_planners/default_experiment_planner.py:285-305).

The torch sub-modules (nn.Conv3d, nn.InstanceNorm3d, ...) exist ONLY as parameter holders so that
`network.apply(InitWeights_He)`, `state_dict()/load_state_dict()`, DDP wrapping and optimizers see the reference's
names and shapes.  Compute never goes through them: `forward` runs a fixed schedule of hand-written HIP kernels
(libnnuzoo_hip.so) on channels-last fp16 activations with fp32 accumulation - the numerics of the reference's
autocast step (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainer.py:1128-1139) - inside ONE
autograd.Function whose backward is an explicit reverse schedule.  CPU tensors raise: there is no fallback.

Data layout in HBM (per resolution level s, C_s features):
  raw_b   [N, V_s, C_s]  fp16  conv output before norm, one per conv block (saved for backward)
  nstat_b [N, C_s, 4]    fp32  the block's InstanceNorm table {mean, rstd, scale, shift}
  cat_s   [N, V_s, 2C_s] fp16  decoder input: [..., :C_s] = transposed-conv output, [..., C_s:] = encoder skip
                               (the last encoder block of the level writes straight into the second half: torch.cat is
                               never materialised)
Round 4 - consumer-side norm + activation ("conv + norm + act fused"): the ACTIVATED tensors do not exist.  Every consumer
of a block (the next convolution, the stride-2 convolution of the next stage, the transposed convolution, the segmentation
head, and in the backward the weight-gradient kernels) reads raw_b and applies lrelu(x * scale + shift) from nstat_b while
it stages its operand (hip_ops.InNorm; csrc/conv_fprop.hip ConvDev::in_tab).  The skip half of cat_s therefore holds the RAW
output of the level's last encoder block.  NNZ_CONSUMER_NORM=0 keeps the separate apply pass and the `act` tensors (A/B
runs; both forms give the same bits, tests/test_plain_unet_gpu.py).
"""
from __future__ import annotations

import contextlib
import os
from typing import List, Optional, Sequence, Tuple, Type, Union

import numpy as np
import torch
from torch import nn

from .. import conv_plan as cp
from .. import hip_ops as ops
from ..hip_ops import PreparedTable


# ------------------------------------------------------------------------------------------------------------------
# parameter-holder modules (names mirror dynamic_network_architectures.building_blocks)
# ------------------------------------------------------------------------------------------------------------------
def _nd(conv_op) -> int:
    return {nn.Conv2d: 2, nn.Conv3d: 3}[conv_op]


def _nlist(v, nd: int) -> List[int]:
    return [int(v)] * nd if isinstance(v, int) else [int(i) for i in v]


def _to3(v: Sequence[int], fill: int = 1) -> Tuple[int, int, int]:
    """2-D layer geometry as a depth-1 volume: (h, w) -> (fill, h, w)"""
    v = tuple(int(i) for i in v)
    return v if len(v) == 3 else (fill, *v)


def _no_direct_call(self, *a, **k):
    raise RuntimeError(f"{type(self).__name__} is a parameter holder of nnuzoo_amd.PlainConvUNet; call the network")


class ConvDropoutNormReLU(nn.Module):
    def __init__(self, conv_op, cin, cout, kernel_size, stride, conv_bias, norm_op, norm_op_kwargs, nonlin,
                 nonlin_kwargs):
        super().__init__()
        self.input_channels, self.output_channels = cin, cout
        nd = _nd(conv_op)
        ks, st = _nlist(kernel_size, nd), _nlist(stride, nd)
        self.stride, self.kernel_size = st, ks
        self.conv = conv_op(cin, cout, ks, st, padding=[(k - 1) // 2 for k in ks], dilation=1, bias=conv_bias)
        mods = [self.conv]
        self.norm = norm_op(cout, **(norm_op_kwargs or {}))
        mods.append(self.norm)
        self.nonlin = nonlin(**(nonlin_kwargs or {}))
        mods.append(self.nonlin)
        self.all_modules = nn.Sequential(*mods)

    forward = _no_direct_call


class StackedConvBlocks(nn.Module):
    def __init__(self, num_convs, conv_op, cin, cout, kernel_size, initial_stride, conv_bias, norm_op,
                 norm_op_kwargs, nonlin, nonlin_kwargs):
        super().__init__()
        if not isinstance(cout, (tuple, list)):
            cout = [cout] * num_convs
        blocks = [ConvDropoutNormReLU(conv_op, cin, cout[0], kernel_size, initial_stride, conv_bias, norm_op,
                                      norm_op_kwargs, nonlin, nonlin_kwargs)]
        for i in range(1, num_convs):
            blocks.append(ConvDropoutNormReLU(conv_op, cout[i - 1], cout[i], kernel_size, 1, conv_bias, norm_op,
                                              norm_op_kwargs, nonlin, nonlin_kwargs))
        self.convs = nn.Sequential(*blocks)
        self.output_channels = cout[-1]

    forward = _no_direct_call


class PlainConvEncoder(nn.Module):
    def __init__(self, input_channels, n_stages, features_per_stage, conv_op, kernel_sizes, strides, n_conv_per_stage,
                 conv_bias, norm_op, norm_op_kwargs, nonlin, nonlin_kwargs):
        super().__init__()
        stages = []
        cin = input_channels
        for s in range(n_stages):
            stages.append(nn.Sequential(StackedConvBlocks(n_conv_per_stage[s], conv_op, cin, features_per_stage[s],
                                                          kernel_sizes[s], strides[s], conv_bias, norm_op,
                                                          norm_op_kwargs, nonlin, nonlin_kwargs)))
            cin = features_per_stage[s]
        self.stages = nn.Sequential(*stages)
        self.output_channels = list(features_per_stage)
        self.strides = [_nlist(s, _nd(conv_op)) for s in strides]
        self.return_skips = True
        self.conv_op, self.norm_op, self.norm_op_kwargs = conv_op, norm_op, norm_op_kwargs
        self.nonlin, self.nonlin_kwargs, self.conv_bias = nonlin, nonlin_kwargs, conv_bias
        self.kernel_sizes = kernel_sizes

    forward = _no_direct_call


class UNetDecoder(nn.Module):
    def __init__(self, encoder: PlainConvEncoder, num_classes, n_conv_per_stage, deep_supervision):
        super().__init__()
        self.deep_supervision = deep_supervision
        self.encoder = encoder
        self.num_classes = num_classes
        n_enc = len(encoder.output_channels)
        if isinstance(n_conv_per_stage, int):
            n_conv_per_stage = [n_conv_per_stage] * (n_enc - 1)
        assert len(n_conv_per_stage) == n_enc - 1
        transp_op = {nn.Conv3d: nn.ConvTranspose3d, nn.Conv2d: nn.ConvTranspose2d}[encoder.conv_op]
        stages, transpconvs, seg_layers = [], [], []
        for s in range(1, n_enc):
            below, skip = encoder.output_channels[-s], encoder.output_channels[-(s + 1)]
            st = encoder.strides[-s]
            transpconvs.append(transp_op(below, skip, st, st, bias=encoder.conv_bias))
            stages.append(StackedConvBlocks(n_conv_per_stage[s - 1], encoder.conv_op, 2 * skip, skip,
                                            encoder.kernel_sizes[-(s + 1)], 1, encoder.conv_bias, encoder.norm_op,
                                            encoder.norm_op_kwargs, encoder.nonlin, encoder.nonlin_kwargs))
            seg_layers.append(encoder.conv_op(skip, num_classes, 1, 1, 0, bias=True))
        self.stages = nn.ModuleList(stages)
        self.transpconvs = nn.ModuleList(transpconvs)
        self.seg_layers = nn.ModuleList(seg_layers)

    forward = _no_direct_call


# ------------------------------------------------------------------------------------------------------------------
# execution plan
# ------------------------------------------------------------------------------------------------------------------
class _Block:
    """One conv -> InstanceNorm -> LeakyReLU block of the schedule.  Geometry is always 3-D: a 2-D layer is the
    depth-1 volume with kernel (1, kh, kw) and stride (1, sh, sw)."""

    def __init__(self, holder: ConvDropoutNormReLU, N, in_dims, first: bool):
        self.h = holder
        self.ks, self.stride = _to3(holder.kernel_size), _to3(holder.stride)
        self.nk = self.ks[0] * self.ks[1] * self.ks[2]
        self.cin_w, self.cout = holder.input_channels, holder.output_channels
        # the network's first conv: Cin = 1, k3^3, 32 features runs the dedicated VALU/MFMA stem kernels (the input
        # stays fp32 NCDHW); any other first conv (2-D, several modalities, other kernels) pads the input channels to
        # 32 zeros-filled fp16 channels and runs the ordinary MFMA path.
        self.stem = first and self.cin_w == 1 and self.ks == (3, 3, 3) and self.stride == (1, 1, 1) and self.cout == 32
        self.padded = first and not self.stem
        self.cin = 32 * ((self.cin_w + 31) // 32) if self.padded else self.cin_w
        self.in_dims = tuple(in_dims)
        self.out_dims = cp.conv_out_dims(in_dims, self.ks, self.stride)
        self.N = N
        self.V = int(np.prod(self.out_dims))
        self.eps = float(holder.norm.eps)
        self.slope = float(getattr(holder.nonlin, "negative_slope", 0.01))
        # filled by the planner
        self.x_ld = self.cin      # channel stride of the input activation
        self.y_ld = self.cout     # channel stride of the output activation (2C when written into a cat buffer)
        self.fwd = self.dgrad = self.wgrad = None
        self.zero_dx = cp.dgrad_uncovered(self.ks, self.stride)
        self.raw_ld = self.cout   # channel stride of the block's raw conv output (y_ld when consumers normalise on the fly)
        self.wgrad_flipped = False

    def prepare(self, consumer_norm: bool):
        # consumer-side norm: the raw output IS what later kernels read, so a skip block's convolution writes into the cat buffer
        self.raw_ld = self.y_ld if consumer_norm else self.cout
        if not self.stem:
            self.fwd = PreparedTable(cp.conv_forward(self.N, self.in_dims, self.cin, self.cout, ks=self.ks,
                                                     stride=self.stride, ldi=self.x_ld, ldo=self.raw_ld))
            # stride 1: operand roles exchanged (conv_plan.conv_wgrad_flipped) so that the layer input is the halo-free plain
            # operand of the weight-gradient kernel - with consumer-side norm every voxel of the RAW input is then
            # normalised once per tile instead of 2.3 times.  Both modes use the same form (same summation order: the two
            # schedules stay bit-identical).
            self.wgrad_flipped = self.stride == (1, 1, 1) and not self.padded
            if self.wgrad_flipped:
                self.wgrad = PreparedTable(cp.conv_wgrad_flipped(self.N, self.in_dims, self.cin, self.cout, ks=self.ks,
                                                                 ldx=self.x_ld, lddy=self.cout))
            else:
                self.wgrad = PreparedTable(cp.conv_wgrad(self.N, self.in_dims, self.cin, self.cout, ks=self.ks,
                                                         stride=self.stride, ldx=self.x_ld, lddy=self.cout))
        if not self.stem and not self.padded:
            # dgrad: in = d(raw) [ld cout] -> out = d(input activation) [ld x_ld]
            self.dgrad = PreparedTable(cp.conv_dgrad(self.N, self.in_dims, self.cin, self.cout, ks=self.ks,
                                                     stride=self.stride, ldi=self.cout, ldo=self.x_ld))
            self.dgrad_acc = self.dgrad.with_accumulate(True)


class _Up:
    def __init__(self, tconv: nn.Module, N, in_dims, cin, cout, ldo):
        self.m = tconv
        self.N, self.in_dims, self.cin, self.cout = N, tuple(in_dims), cin, cout
        st = self.stride = _to3(tconv.stride)
        assert _to3(tconv.kernel_size) == st
        self.nk = st[0] * st[1] * st[2]
        self.out_dims = tuple(st[a] * in_dims[a] for a in range(3))
        self.fwd = PreparedTable(cp.convT_forward(N, in_dims, cin, cout, ldi=cin, ldo=ldo, stride=st))
        self.dgrad = PreparedTable(cp.convT_dgrad(N, in_dims, cin, cout, ldi=ldo, ldo=cin, stride=st))
        self.wgrad = PreparedTable(cp.convT_wgrad(N, in_dims, cin, cout, lddout=ldo, ldin=cin, stride=st))
        self.ldo = ldo
        # full-resolution stages: the dedicated HBM-bound kernel (weights resident in LDS); deeper stages: tap-table path
        w_ok = tconv.weight.dtype == torch.float32 and tconv.weight.is_contiguous() and os.environ.get("NNZ_CONVT", "1") != "0"
        self.native_fwd = w_ok and ops.convT_supported(cin, cout, st, False)
        self.native_dgrad = w_ok and ops.convT_supported(cin, cout, st, True)


class _Plan:
    def __init__(self, net: "PlainConvUNet", N: int, dims: Tuple[int, int, int]):
        enc, dec = net.encoder, net.decoder
        self.N, self.dims = N, tuple(dims)
        self.consumer_norm = bool(net.consumer_norm)
        S = len(enc.stages)
        self.S = S
        feats = enc.output_channels
        # ---- encoder blocks
        self.enc_blocks: List[List[_Block]] = []
        d = tuple(dims)
        self.level_dims = []
        for s in range(S):
            blocks = []
            for i, h in enumerate(enc.stages[s][0].convs):
                b = _Block(h, N, d, first=(s == 0 and i == 0))
                d = b.out_dims
                blocks.append(b)
            self.level_dims.append(d)
            self.enc_blocks.append(blocks)
        # skip of level s (< S-1) lives in cat_s[..., C:2C]
        for s in range(S - 1):
            self.enc_blocks[s][-1].y_ld = 2 * feats[s]
        # ---- decoder (index j = 0 is the deepest decoder stage, like decoder.stages)
        self.ups: List[_Up] = []
        self.dec_blocks: List[List[_Block]] = []
        for j in range(S - 1):
            lvl = S - 2 - j
            below, skip = feats[lvl + 1], feats[lvl]
            up = _Up(dec.transpconvs[j], N, self.level_dims[lvl + 1], below, skip, ldo=2 * skip)
            if up.out_dims != tuple(self.level_dims[lvl]):
                raise ValueError(f"nnuzoo_amd.PlainConvUNet: patch size {tuple(dims)} is not divisible by the network's "
                                 f"strides (level {lvl}: {up.out_dims} after upsampling vs {self.level_dims[lvl]})")
            self.ups.append(up)
            blocks = []
            for i, h in enumerate(dec.stages[j].convs):
                b = _Block(h, N, self.level_dims[lvl], first=False)
                if i == 0:
                    b.x_ld = 2 * skip
                blocks.append(b)
            self.dec_blocks.append(blocks)
        # input strides of encoder blocks that read a skip stored inside a cat buffer
        for s in range(1, S):
            self.enc_blocks[s][0].x_ld = 2 * feats[s - 1] if s - 1 < S - 1 else feats[s - 1]
        for s in range(S):
            for i, b in enumerate(self.enc_blocks[s]):
                if i > 0:
                    b.x_ld = self.enc_blocks[s][i - 1].y_ld
        self.all_blocks: List[_Block] = [b for blocks in self.enc_blocks + self.dec_blocks for b in blocks]
        so = 0
        for b in self.all_blocks:
            if self.consumer_norm and b.stem and b.y_ld != b.cout:
                raise NotImplementedError("nnuzoo_amd.PlainConvUNet: a one-conv first stage (the stem kernel writing into a "
                                          "cat buffer) needs NNZ_CONSUMER_NORM=0")
            b.prepare(self.consumer_norm)
            b.stats_off, so = so, so + N * b.cout * 4          # slice of the per-step InstanceNorm tables (nstat)
        self.stats_floats = so
        self.norm_scratch = None                                  # fixed-point accumulators + counter (first use)
        self.norm_capacity = N * max([b.cout for b in self.all_blocks] + [u.cout for u in self.ups])
        # partial-block workspace of the two-stage weight gradients (shared by all layers: they run one after another)
        self.wgrad_ws_floats = max([ops.conv_tap_wgrad_workspace_floats(b.wgrad) for b in self.all_blocks if not b.stem]
                                   + [ops.conv_tap_wgrad_workspace_floats(u.wgrad) for u in self.ups])
        self.wgrad_ws = None
        self.pack_fwd = self.pack_dual = None                     # built on first use (needs device pointers)

    def build_pack_tables(self, dev):
        """persistent packed-weight buffers + device job tables: ONE pack launch per forward and per backward"""
        f16 = torch.float16
        # pack_fwd: only what the dual pack does not cover (a zero-padded first conv: forward form only)
        self.pack_fwd, self.pack_dual = ops.PackJobTable(dev), ops.DualPackTable(dev)
        for b in self.all_blocks:
            if b.stem:
                continue
            w = b.h.conv.weight
            nk = b.nk
            b.wp_fwd = torch.empty(b.cin * b.cout * nk, dtype=f16, device=dev)
            if b.padded:
                # fp32 staging copy of the first conv's weight with the input channels padded to 32 (refreshed from
                # the parameter before every pack; the padding rows stay zero)
                b.w_pad = torch.zeros((b.cout, b.cin, *b.ks), dtype=torch.float32, device=dev)
                b.gw_pad = torch.empty_like(b.w_pad)
                self.pack_fwd.add(b.w_pad, b.wp_fwd, b.fwd, b.cin, b.cout, nk, b.cin * nk, 1)
                continue
            b.wp_dgrad = torch.empty(b.cin * b.cout * nk, dtype=f16, device=dev)
            self.pack_dual.add(w, b.wp_fwd, b.wp_dgrad, b.cin, b.cout, nk, True, b.fwd, b.dgrad)
        for u in self.ups:
            w = u.m.weight
            nk = u.nk
            u.wp_fwd = torch.empty(u.cin * u.cout * nk, dtype=f16, device=dev)
            u.wp_dgrad = torch.empty(u.cin * u.cout * nk, dtype=f16, device=dev)
            self.pack_dual.add(w, u.wp_fwd, u.wp_dgrad, u.cin, u.cout, nk, False, u.fwd, u.dgrad)


_SIDE_STREAMS = {}


def _side_stream(dev) -> "torch.cuda.Stream":
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return st


def _check_supported(net: "PlainConvUNet"):
    enc = net.encoder
    if enc.conv_op not in (nn.Conv2d, nn.Conv3d):
        raise NotImplementedError("nnuzoo_amd.PlainConvUNet: conv_op must be torch.nn.Conv2d or torch.nn.Conv3d")
    norm_cls = {nn.Conv2d: nn.InstanceNorm2d, nn.Conv3d: nn.InstanceNorm3d}[enc.conv_op]
    if not (isinstance(enc.norm_op, type) and issubclass(enc.norm_op, norm_cls)) or \
            not (enc.norm_op_kwargs or {}).get("affine", False):
        raise NotImplementedError(f"nnuzoo_amd.PlainConvUNet: norm_op must be {norm_cls.__name__}(affine=True)")
    if enc.nonlin is not nn.LeakyReLU:
        raise NotImplementedError("nnuzoo_amd.PlainConvUNet: nonlin must be torch.nn.LeakyReLU")
    nd = _nd(enc.conv_op)
    for s, ks in enumerate(enc.kernel_sizes):
        ks, st = _nlist(ks, nd), enc.strides[s]
        if len(ks) != nd or len(st) != nd or any(k not in (1, 3) for k in ks) or any(v not in (1, 2) for v in st):
            raise NotImplementedError("nnuzoo_amd.PlainConvUNet: per-axis kernel sizes 1/3 and strides 1/2 only "
                                      f"(stage {s}: kernel {ks}, stride {st})")
    if any(v != 1 for v in enc.strides[0]):
        raise NotImplementedError("nnuzoo_amd.PlainConvUNet: the first stage must have stride 1 (planner default)")
    feats = enc.output_channels
    if any(f % 32 for f in feats):
        raise NotImplementedError("nnuzoo_amd.PlainConvUNet: features per stage must be multiples of 32")
    for s in range(1, len(feats)):
        if nd == 3 and any(v == 2 for v in enc.strides[s]) and feats[s] % 64:
            raise NotImplementedError("nnuzoo_amd.PlainConvUNet: a strided 3-D stage needs features % 64 == 0")
    if not enc.conv_bias:
        raise NotImplementedError("nnuzoo_amd.PlainConvUNet: conv_bias=True expected (planner default)")


class _UNetFunction(torch.autograd.Function):
    """Whole-network forward/backward as one autograd node (explicit schedule, no per-op autograd graph)."""

    @staticmethod
    def forward(ctx, net: "PlainConvUNet", save: bool, x: torch.Tensor, *params: torch.Tensor):
        outs, saved = net._run_forward(x, save=save)
        ctx.net = net
        ctx.saved = saved
        ctx.set_materialize_grads(False)  # unused deep-supervision outputs arrive as None, not as zero tensors
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        net = ctx.net
        grads = net._run_backward(ctx.saved, gouts)
        ctx.saved = None
        return (None, None, None, *grads)


class PlainConvUNet(nn.Module):
    def __init__(self,
                 input_channels: int,
                 n_stages: int,
                 features_per_stage: Union[int, Sequence[int]],
                 conv_op: Type[nn.Module],
                 kernel_sizes,
                 strides,
                 n_conv_per_stage: Union[int, Sequence[int]],
                 num_classes: int,
                 n_conv_per_stage_decoder: Union[int, Sequence[int]],
                 conv_bias: bool = False,
                 norm_op=None,
                 norm_op_kwargs: dict = None,
                 dropout_op=None,
                 dropout_op_kwargs: dict = None,
                 nonlin=None,
                 nonlin_kwargs: dict = None,
                 deep_supervision: bool = False,
                 nonlin_first: bool = False):
        super().__init__()
        if isinstance(n_conv_per_stage, int):
            n_conv_per_stage = [n_conv_per_stage] * n_stages
        if isinstance(n_conv_per_stage_decoder, int):
            n_conv_per_stage_decoder = [n_conv_per_stage_decoder] * (n_stages - 1)
        if isinstance(features_per_stage, int):
            features_per_stage = [features_per_stage] * n_stages
        if isinstance(kernel_sizes, int):
            kernel_sizes = [kernel_sizes] * n_stages
        if isinstance(strides, int):
            strides = [strides] * n_stages
        if dropout_op is not None:
            raise NotImplementedError("nnuzoo_amd.PlainConvUNet: dropout_op must be None (planner default)")
        if nonlin_first:
            raise NotImplementedError("nnuzoo_amd.PlainConvUNet: nonlin_first=False only")
        self.input_channels = input_channels
        self.num_classes = num_classes
        self.encoder = PlainConvEncoder(input_channels, n_stages, list(features_per_stage), conv_op, list(kernel_sizes),
                                        list(strides), list(n_conv_per_stage), conv_bias, norm_op, norm_op_kwargs,
                                        nonlin, nonlin_kwargs)
        self.decoder = UNetDecoder(self.encoder, num_classes, list(n_conv_per_stage_decoder), deep_supervision)
        _check_supported(self)
        self._nd = _nd(conv_op)
        self._plans = {}
        self._param_list: Optional[List[nn.Parameter]] = None
        self.grad_reducer = None  # set by nnuzoo_amd.ddp.attach_bucketed_allreduce
        # data-gradient launches also close the InstanceNorm-backward reductions of the layer below (see _conv_block_bwd);
        # NNZ_FUSE_NORM_REDUCE=0 keeps the separate reducing launches (A/B runs, tests)
        self.fuse_norm_reduce = os.environ.get("NNZ_FUSE_NORM_REDUCE", "1") != "0"
        # NNZ_WGRAD_SIDE_MAXV > 0: the weight gradients of the conv blocks with at most that many output voxels per sample run on a
        # SIDE stream (they depend on the block's norm-backward output only, not on the data-gradient chain): the under-filled launches
        # of the deep levels overlap the chain instead of extending it; captured as a parallel branch when the step is a hipGraph.
        # Not with a gradient reducer (N > 1: the hand-over points expect the block's gradients in stream order).
        self.wgrad_side_maxv = int(os.environ.get("NNZ_WGRAD_SIDE_MAXV", "0"))
        # consumers normalise + activate raw conv outputs while staging them; no apply pass, no activated tensors (module doc)
        self.consumer_norm = os.environ.get("NNZ_CONSUMER_NORM", "1") != "0"

    # reference API: `network.apply(network.initialize)` (get_network_from_plans.py:59-60)
    @staticmethod
    def initialize(module):
        from ..utilities.network_initialization import InitWeights_He
        InitWeights_He(1e-2)(module)

    # ---- plumbing ------------------------------------------------------------------------------------------------
    def _params(self) -> List[nn.Parameter]:
        if self._param_list is None:
            self._param_list = list(self.parameters())
        return self._param_list

    def _plan(self, N, dims) -> _Plan:
        key = (N, tuple(dims))
        if key not in self._plans:
            self._plans[key] = _Plan(self, N, dims)
        return self._plans[key]

    def forward(self, x: torch.Tensor):
        if not x.is_cuda:
            raise RuntimeError("nnuzoo_amd.PlainConvUNet runs on MI355X through libnnuzoo_hip.so only; got a CPU "
                               "tensor and there is deliberately no CPU fallback (see oracle/ for the test-only "
                               "CPU restatement)")
        nd = _nd(self.encoder.conv_op)
        if x.dim() != nd + 2 or x.shape[1] != self.input_channels:
            raise ValueError(f"expected (B, {self.input_channels}, {'D, H, W' if nd == 3 else 'H, W'}) input, got "
                             f"{tuple(x.shape)}")
        x = x.float().contiguous()
        params = self._params()
        save = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        outs = _UNetFunction.apply(self, save, x, *params)
        if self.decoder.deep_supervision:
            return list(outs)
        return outs[0]

    # ---- forward schedule ----------------------------------------------------------------------------------------
    def _conv_block_fwd(self, b: _Block, x_in: torch.Tensor, x_norm, out_buf: torch.Tensor, dev, stats_all):
        """One block.  x_in / x_norm: the input operand and its hip_ops.InNorm (None: x_in is used as it is).  out_buf: where
        consumers will read the block from (a cat-buffer slice for skip blocks).  Returns (raw, stats, out, out_norm): the raw
        conv output with its table, and the operand + InNorm later kernels take - (raw, table) with consumer-side norm, the
        materialised activation otherwise."""
        h = b.h
        cn = self._plan_cn
        raw = out_buf if cn else torch.empty((b.N, b.V, b.cout), dtype=torch.float16, device=dev)
        # the block's InstanceNorm table {mean, rstd, scale, shift} per (sample, channel): written by the last workgroup
        # of the launch that produces the statistics (deterministic fixed-point sums, csrc/common.hpp)
        stats = stats_all[b.stats_off:b.stats_off + b.N * b.cout * 4].view(b.N, b.cout, 4)
        scratch = self._scratch
        if b.stem:
            ops.stem_forward(x_in, h.conv.weight, h.conv.bias, raw, (b.N, *b.in_dims), b.cout)
            ops.instnorm_stats_det(raw, b.N, b.V, b.cout, b.cout, scratch, h.norm.weight, h.norm.bias, b.eps, nstat=stats)
        else:
            # the convolution's epilogue accumulates the InstanceNorm statistics of the tile it just produced
            ops.conv_tap_forward_norm(b.fwd, self._padded_input(b, x_in) if b.padded else x_in, b.wp_fwd, h.conv.bias,
                                      raw, scratch, h.norm.weight, h.norm.bias, b.eps, stats, workspace=self._ws,
                                      innorm=x_norm)
        if cn:
            return raw, stats, raw, ops.InNorm(stats, b.slope)
        ops.instnorm_lrelu_apply_tab(raw, stats, out_buf, b.N, b.V, b.cout, b.cout, b.y_ld, b.slope)
        return raw, stats, out_buf, None

    @staticmethod
    def _padded_input(b: _Block, x: torch.Tensor) -> torch.Tensor:
        """fp32 (N, C, *dims) network input -> channels-last fp16 [N, V, 32k] with zero-filled padding channels
        (layout plumbing for first convs the dedicated stem kernels do not cover)"""
        xc = torch.zeros((b.N, int(np.prod(b.in_dims)), b.cin), dtype=torch.float16, device=x.device)
        xc[:, :, :b.cin_w] = x.reshape(b.N, b.cin_w, -1).transpose(1, 2)
        return xc

    def _run_forward(self, x: torch.Tensor, save: bool):
        N = x.shape[0]
        dims = _to3(x.shape[2:])
        plan = self._plan(N, dims)
        dev = x.device
        S = plan.S
        feats = self.encoder.output_channels
        K = self.num_classes
        f16 = torch.float16
        if plan.pack_fwd is None:
            plan.build_pack_tables(dev)
        b0 = plan.enc_blocks[0][0]
        if b0.padded:
            b0.w_pad[:, :b0.cin_w].copy_(b0.h.conv.weight.detach().reshape(b0.cout, b0.cin_w, *b0.ks))
        if plan.pack_fwd.jobs:
            plan.pack_fwd.run()
        plan.pack_dual.run()     # forward AND data-gradient forms from one read of the parameters (one launch per step)
        if plan.norm_scratch is None:
            plan.norm_scratch = ops.NormScratch(dev, plan.norm_capacity)
        self._scratch = plan.norm_scratch
        if plan.wgrad_ws is None:   # fp32 workspace shared by the weight gradients and the split-K layers (stream order)
            plan.wgrad_ws = torch.empty(plan.wgrad_ws_floats, dtype=torch.float32, device=dev)
        self._ws = plan.wgrad_ws
        stats_all = torch.empty(plan.stats_floats, dtype=torch.float32, device=dev)
        cats = [torch.empty((N, int(np.prod(plan.level_dims[s])), 2 * feats[s]), dtype=f16, device=dev)
                for s in range(S - 1)]
        rec = {"x": x, "plan": plan, "cats": cats, "enc": [], "dec": [], "heads": []}
        self._plan_cn = plan.consumer_norm
        # cur / cur_norm: the operand the next kernel reads and its consumer-side norm (None: used as it is)
        cur, cur_norm = x, None
        for s in range(S):
            stage_rec = []
            for i, b in enumerate(plan.enc_blocks[s]):
                last = i == len(plan.enc_blocks[s]) - 1
                if last and s < S - 1:
                    buf = cats[s][:, :, feats[s]:]
                else:
                    buf = torch.empty((N, b.V, b.cout), dtype=f16, device=dev)
                raw, stats, out, out_norm = self._conv_block_fwd(b, cur, cur_norm, buf, dev, stats_all)
                stage_rec.append((cur, raw, stats, out, cur_norm, out_norm))
                cur, cur_norm = out, out_norm
            rec["enc"].append(stage_rec)
        outs = [None] * (S - 1)
        lres, lres_norm = cur, cur_norm
        for j in range(S - 1):
            lvl = S - 2 - j
            up = plan.ups[j]
            if up.native_fwd:
                ops.convT_forward(lres, up.m.weight, up.m.bias, cats[lvl], up.N, up.in_dims, up.cin, up.cout, up.stride,
                                  up.cin, up.ldo, innorm=lres_norm)
            else:
                ops.conv_tap_forward(up.fwd, lres, up.wp_fwd, up.m.bias, cats[lvl], innorm=lres_norm)
            # decoder input = [transposed conv | skip]: with consumer-side norm the skip half is the raw output of the level's
            # last encoder block, normalised by the reader (channels below C pass unchanged)
            skip_norm = rec["enc"][lvl][-1][5]
            cur = cats[lvl]
            cur_norm = ops.InNorm(skip_norm.tab, skip_norm.slope, c0=feats[lvl]) if skip_norm is not None else None
            stage_rec = []
            for b in plan.dec_blocks[j]:
                buf = torch.empty((N, b.V, b.cout), dtype=f16, device=dev)
                raw, stats, out, out_norm = self._conv_block_fwd(b, cur, cur_norm, buf, dev, stats_all)
                stage_rec.append((cur, raw, stats, out, cur_norm, out_norm))
                cur, cur_norm = out, out_norm
            rec["dec"].append((lres, stage_rec, lres_norm))
            if self.decoder.deep_supervision or j == S - 2:
                seg = self.decoder.seg_layers[j]
                V = int(np.prod(plan.level_dims[lvl]))
                logits = torch.empty((N, K, *plan.level_dims[lvl][3 - self._nd:]), dtype=f16, device=dev)
                ops.head_forward(cur, seg.weight, seg.bias, logits, N, V, feats[lvl], K, feats[lvl], innorm=cur_norm)
                outs[lvl] = logits
            lres, lres_norm = cur, cur_norm
        out_list = [o for o in outs if o is not None]  # highest resolution first
        rec["out_levels"] = [lvl for lvl, o in enumerate(outs) if o is not None]
        return out_list, (rec if save else None)

    # ---- backward schedule ---------------------------------------------------------------------------------------
    def _conv_block_bwd(self, b: _Block, recd, g_act: torch.Tensor, g_ld: int, grads: dict, dx_out, dx_acc: bool, dev,
                        below=None):
        """g_act: gradient wrt the block's activated output (channel stride g_ld).  Writes the gradient wrt the
        block input into dx_out (None for the stem) and the parameter gradients into `grads`.

        below = (block, record) of the conv block whose ACTIVATION dx_out is the gradient of (None: a tensor without a
        norm below it, or a gradient that other launches still add to).  The data-gradient launch then also closes that
        block's InstanceNorm-backward reductions in its epilogue (csrc/conv_fprop.hip, ConvDev::bx) and the block's own
        call of this method runs the apply launch only."""
        x_in, raw, stats, _, x_norm, _ = recd
        h = b.h
        red = self._red_all[b.stats_off // 2:b.stats_off // 2 + b.N * b.cout * 2].view(b.N, b.cout, 2)
        draw = torch.empty((b.N, b.V, b.cout), dtype=torch.float16, device=dev)
        pre = self._prereduced.pop(id(b), None)
        if pre is not None:
            gnw, gnb = pre
            ops.instnorm_lrelu_bwd_apply_tab(raw, g_act, stats, red, draw, b.N, b.V, b.cout, b.raw_ld, g_ld, b.cout, b.slope)
        else:
            gnw, gnb = self._galloc(h.norm.weight), self._galloc(h.norm.bias)
            ops.instnorm_lrelu_bwd_tab(raw, g_act, stats, self._scratch, red, draw, b.N, b.V, b.cout, b.raw_ld, g_ld, b.cout,
                                       b.slope, dgamma=gnw, dbeta=gnb)
        grads[h.norm.weight], grads[h.norm.bias] = gnw, gnb
        # the bias of a conv followed by InstanceNorm has an identically zero gradient (mean removal)
        grads[h.conv.bias] = self._galloc(h.conv.bias)  # arena is zero-initialised
        gw = self._galloc(h.conv.weight)
        if b.stem:
            ops.stem_wgrad(x_in, draw, gw, (b.N, *b.in_dims), b.cout, scratch=self._scratch)
        else:
            nk = b.nk
            # dW[t][cin][cout] -> torch layout (cout, cin, *k): a = cin (stride nk), b = cout (stride cin*nk), t stride 1
            if b.padded:
                ops.conv_tap_wgrad_to_grad(b.wgrad, self._padded_input(b, x_in), draw, self._wgrad_ws, b.gw_pad, nk,
                                           b.cin * nk, 1)
                gw.copy_(b.gw_pad[:, :b.cin_w].reshape(gw.shape))
            else:
                side = self._side if (self._side is not None and b.V <= self.wgrad_side_maxv) else None
                if side is not None:
                    side.wait_stream(torch.cuda.current_stream(dev))      # draw (and the arena's zero fill) are complete
                    draw.record_stream(side)
                    ws = self._side_ws
                else:
                    ws = self._wgrad_ws
                with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                    if b.wgrad_flipped:   # dW[t][cout][cin]: a = cout (stride cin*nk), b = cin (stride nk)
                        ops.conv_tap_wgrad_to_grad(b.wgrad, draw, x_in, ws, gw, b.cin * nk, nk, 1, plain_norm=x_norm)
                    else:
                        ops.conv_tap_wgrad_to_grad(b.wgrad, x_in, draw, ws, gw, nk, b.cin * nk, 1, boxed_norm=x_norm)
                if side is not None:
                    self._side_used = True
                if b.zero_dx and not dx_acc:
                    dx_out.zero_()  # k1 s2 axes: odd input positions are outside every output's footprint
                pt = b.dgrad_acc if dx_acc else b.dgrad
                # (not on the <= 8^3 levels: there the data gradient runs split-K, which has no fused epilogue, and the
                # separate reducing launch over a few hundred voxels costs nothing)
                if below is not None and self.fuse_norm_reduce and not b.zero_dx and below[0].cout == b.cin \
                        and below[0].V >= 4096 and b.N * max(b.V * b.cout, below[0].V * pt.desc.ldo) < 2 ** 31:
                    pb, prec = below
                    pred = self._red_all[pb.stats_off // 2:pb.stats_off // 2 + pb.N * pb.cout * 2].view(pb.N, pb.cout, 2)
                    pgw, pgb = self._galloc(pb.h.norm.weight), self._galloc(pb.h.norm.bias)
                    ops.conv_tap_dgrad_normred(pt, draw, b.wp_dgrad, dx_out, prec[1], pb.raw_ld, prec[2], pb.slope,
                                               self._scratch, pred, pgw, pgb)
                    self._prereduced[id(pb)] = (pgw, pgb)
                else:
                    ops.conv_tap_forward(pt, draw, b.wp_dgrad, None, dx_out, workspace=self._wgrad_ws)
        grads[h.conv.weight] = gw
        if self.grad_reducer is not None:
            # hand-over point per conv block (finer than per stage: the 320-channel decoder stage alone is 37 MB)
            self.grad_reducer.stage_done_arena(self._arena, self._arena_off)

    def _galloc(self, like: torch.Tensor, unused: bool = False) -> torch.Tensor:
        """Gradient storage comes from ONE flat fp32 arena, filled in backward-completion order: the data-parallel
        reducer all-reduces contiguous slices of it in place (no flatten / unflatten copies) and the fused optimizer
        (training/fused_sgd.py) reads it directly.  unused=True marks a parameter that took no part in the loss (the
        seg head of a deep-supervision output with weight 0, deep_supervision.py:30): its slice stays zero so that the
        arena layout and the reducer's slices are the same on every rank, but autograd receives None for it and the
        fused optimizer skips it - like torch, which leaves a parameter without .grad untouched (no weight decay,
        no momentum update)."""
        n = like.numel()
        off = self._arena_off
        self._arena_off = off + n
        self._arena_trace.append((like, off))
        if unused:
            self._arena_unused.add(id(like))
        return self._arena[off:off + n].view(like.shape)

    # ---- gradient arena of the most recent backward (fused optimizer) ---------------------------------------------
    def grad_arena(self) -> Optional[torch.Tensor]:
        return getattr(self, "_last_arena", None)

    def grad_arena_layout(self):
        """[(parameter, element offset)] in arena order; fixed by the schedule (identical every step)"""
        return list(self._arena_layout)

    def grad_arena_unused(self):
        """ids of the parameters whose arena slice carries no gradient in the most recent backward"""
        return frozenset(getattr(self, "_last_unused", ()))

    def release_grad_arena(self):
        self._last_arena = None

    def _run_backward(self, rec, gouts):
        plan: _Plan = rec["plan"]
        N, S = plan.N, plan.S
        feats = self.encoder.output_channels
        K = self.num_classes
        dev = rec["x"].device
        f16 = torch.float16
        grads = {}
        self._arena = torch.zeros(sum(p.numel() for p in self._params()), dtype=torch.float32, device=dev)
        self._arena_off = 0
        self._arena_trace = []
        self._arena_unused = set()
        self._red_all = torch.empty(plan.stats_floats // 2, dtype=torch.float32, device=dev)
        self._prereduced = {}
        self._scratch = plan.norm_scratch
        if plan.wgrad_ws is None:
            plan.wgrad_ws = torch.empty(plan.wgrad_ws_floats, dtype=torch.float32, device=dev)
        self._wgrad_ws = plan.wgrad_ws
        self._side, self._side_used = None, False
        if self.wgrad_side_maxv > 0 and self.grad_reducer is None and dev.type == "cuda":
            self._side = _side_stream(dev)
            if getattr(plan, "wgrad_ws_side", None) is None:      # its own workspace: the main stream's split-K launches use the other
                plan.wgrad_ws_side = torch.empty(plan.wgrad_ws_floats, dtype=torch.float32, device=dev)
            self._side_ws = plan.wgrad_ws_side
        gout_by_level = {lvl: g for lvl, g in zip(rec["out_levels"], gouts)}
        g_cur = None  # gradient wrt the current decoder stage output (act), contiguous [N, V, C]
        g_cats = [None] * (S - 1)
        for j in range(S - 2, -1, -1):
            lvl = S - 2 - j
            C_ = feats[lvl]
            V = int(np.prod(plan.level_dims[lvl]))
            lres, stage_rec, lres_norm = rec["dec"][j]
            out_act, out_norm = stage_rec[-1][3], stage_rec[-1][5]
            seg = self.decoder.seg_layers[j]
            g = gout_by_level.get(lvl)
            if g_cur is None:
                g_cur = torch.empty((N, V, C_), dtype=f16, device=dev)
                have = False
            else:
                have = True
            if g is not None:
                g = g.contiguous()
                if g.dtype != f16:
                    g = g.to(f16)
                gw = self._galloc(seg.weight)
                gb = self._galloc(seg.bias)
                ops.head_wgrad(out_act, g, gw, gb, N, V, C_, K, C_, scratch=self._scratch, innorm=out_norm)
                ops.head_dgrad(g, seg.weight, g_cur, N, V, C_, K, C_, accumulate=have)
                grads[seg.weight], grads[seg.bias] = gw, gb
            else:
                if not have:
                    g_cur.zero_()
                if self.decoder.deep_supervision or j == S - 2:
                    grads[seg.weight] = self._galloc(seg.weight, unused=True)
                    grads[seg.bias] = self._galloc(seg.bias, unused=True)
            # conv blocks of the stage, last to first
            blocks = plan.dec_blocks[j]
            g_act, g_ld = g_cur, C_
            for i in range(len(blocks) - 1, -1, -1):
                b = blocks[i]
                dx = torch.empty((N, V, b.cin), dtype=f16, device=dev)
                self._conv_block_bwd(b, stage_rec[i], g_act, g_ld, grads, dx, False, dev,
                                     below=(blocks[i - 1], stage_rec[i - 1]) if i > 0 else None)
                g_act, g_ld = dx, b.cin
            g_cats[lvl] = g_act  # [N, V, 2C]: [..., :C] -> transposed conv, [..., C:] -> encoder skip
            # transposed conv backward
            up = plan.ups[j]
            g_up = g_act  # channel slice [:C] of the cat gradient, ld = 2C
            Vb = int(np.prod(up.in_dims))
            nk = up.nk
            gw = self._galloc(up.m.weight)
            # dW[t][cout][cin] -> torch layout (cin, cout, *k): a = cout (stride nk), b = cin (stride cout*nk)
            ops.conv_tap_wgrad_to_grad(up.wgrad, g_up, lres, self._wgrad_ws, gw, nk, up.cout * nk, 1, plain_norm=lres_norm)
            grads[up.m.weight] = gw
            if up.m.bias is not None:
                st = torch.empty((N, up.cout, 2), dtype=torch.float32, device=dev)
                ops.instnorm_stats_det(g_up, N, V, up.cout, 2 * up.cout, self._scratch, sums=st)
                gub = self._galloc(up.m.bias)
                gub.copy_(st[:, :, 0].sum(0))
                grads[up.m.bias] = gub
            g_below = torch.empty((N, Vb, up.cin), dtype=f16, device=dev)
            if up.native_dgrad:
                ops.convT_dgrad(g_up, up.m.weight, g_below, up.N, up.in_dims, up.cin, up.cout, up.stride, up.ldo, up.cin)
            else:
                ops.conv_tap_forward(up.dgrad, g_up, up.wp_dgrad, None, g_below)
            g_cur = g_below
            if self.grad_reducer is not None:
                self.grad_reducer.stage_done_arena(self._arena, self._arena_off)
        # encoder, deepest first.  g_cur = gradient wrt the bottleneck activation.
        g_act, g_ld = g_cur, feats[S - 1]
        for s in range(S - 1, -1, -1):
            blocks = plan.enc_blocks[s]
            stage_rec = rec["enc"][s]
            for i in range(len(blocks) - 1, -1, -1):
                b = blocks[i]
                first_of_stage = i == 0
                if b.stem or b.padded:  # the network's first conv: no data gradient
                    self._conv_block_bwd(b, stage_rec[i], g_act, g_ld, grads, None, False, dev)
                    continue
                if first_of_stage:
                    # input is the skip of level s-1 living in cat_{s-1}[..., C:]: accumulate into its gradient
                    Cp = feats[s - 1]
                    dx = g_cats[s - 1][:, :, Cp:]
                    # this launch completes the gradient of the previous stage's output (skip + downsampling path)
                    self._conv_block_bwd(b, stage_rec[i], g_act, g_ld, grads, dx, True, dev,
                                         below=(plan.enc_blocks[s - 1][-1], rec["enc"][s - 1][-1]))
                    g_act, g_ld = dx, 2 * Cp
                else:
                    dx = torch.empty((N, b.V, b.cin), dtype=f16, device=dev)
                    self._conv_block_bwd(b, stage_rec[i], g_act, g_ld, grads, dx, False, dev,
                                         below=(blocks[i - 1], stage_rec[i - 1]))
                    g_act, g_ld = dx, b.cin
            if self.grad_reducer is not None:
                self.grad_reducer.stage_done_arena(self._arena, self._arena_off)
        out = []
        for p in self._params():
            gp = grads.get(p)
            if gp is None:
                gp = self._galloc(p, unused=True)  # e.g. seg heads that were not evaluated (deep supervision off)
            out.append(None if id(p) in self._arena_unused else gp)
        if self.grad_reducer is not None:
            self.grad_reducer.finish_arena(self._arena, self._arena_off)
        if self._side_used:
            torch.cuda.current_stream(dev).wait_stream(self._side)     # join: every weight gradient is complete for the caller
        self._side = None
        self._arena_layout = self._arena_trace
        self._last_arena = self._arena
        self._last_unused = self._arena_unused
        assert not self._prereduced, "a fused norm reduction was produced for a block whose backward never ran"
        self._red_all = self._wgrad_ws = self._arena = self._arena_trace = self._arena_unused = self._scratch = None
        return out

    # reference API (dynamic_network_architectures): used by the planner's VRAM estimate only
    def compute_conv_feature_map_size(self, input_size):
        raise NotImplementedError("planner-side VRAM estimate is out of scope (SURVEY.md §2 row 13)")
