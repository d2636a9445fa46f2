"""Host-side tap tables for the tap-table convolution kernels (nnuzoo_amd/csrc/conv_fprop.hip, conv_wgrad.hip).

Every dense contraction of the PlainConvUNet forward/backward is described as
    out[n, m*OS + ooff_g, co] (+)= sum_{t in g} sum_ci in[n, m*IS + off_t, ci] * W[widx_t][ci][co]
(see include/nnuzoo_hip.h).  The functions below derive the tables from the torch layer definitions the
reference uses (Conv3d k3 p1 s{1,2}; ConvTranspose3d k2 s2, arch kwargs at
/root/reference/nnunetv2/experiment_planning/experiment_planners/default_experiment_planner.py:285-305).
Pure Python, no device access: unit-tested on CPU against a numpy restatement of the tap semantics.
"""
from __future__ import annotations

import ctypes as C
import itertools
from dataclasses import dataclass
from typing import List, Sequence, Tuple

from ._lib import ConvDesc, NNZ_MAX_GROUPS, NNZ_MAX_TAPS


@dataclass
class TapTable:
    """Python mirror of nnz_conv_desc (kept so that tests can interpret it without ctypes)."""
    N: int
    in_dims: Tuple[int, int, int]
    out_dims: Tuple[int, int, int]
    m_dims: Tuple[int, int, int]
    Cin: int
    Cout: int
    ldi: int
    ldo: int
    in_stride: Tuple[int, int, int]
    out_stride: Tuple[int, int, int]
    groups: List[Tuple[Tuple[int, int, int], List[Tuple[Tuple[int, int, int], int]]]]  # (ooff, [(off, widx)])
    accumulate: bool = False

    @property
    def pack_ksel(self) -> List[int]:
        """packed weight slice j (tap position j in table order) <- flat kernel index of the torch parameter"""
        return [t[1] for g in self.groups for t in g[1]]

    @property
    def ntaps(self) -> int:
        return sum(len(g[1]) for g in self.groups)

    def to_desc(self) -> ConvDesc:
        d = ConvDesc()
        d.N = self.N
        for i in range(3):
            d.in_dims[i] = self.in_dims[i]
            d.out_dims[i] = self.out_dims[i]
            d.m_dims[i] = self.m_dims[i]
            d.in_stride[i] = self.in_stride[i]
            d.out_stride[i] = self.out_stride[i]
        d.Cin, d.Cout, d.ldi, d.ldo = self.Cin, self.Cout, self.ldi, self.ldo
        offs = [t[0] for g in self.groups for t in g[1]]
        lo = [min(o[a] for o in offs) for a in range(3)]
        hi = [max(o[a] for o in offs) for a in range(3)]
        for a in range(3):
            d.lo[a] = lo[a]
            d.ext[a] = hi[a] - lo[a]
        assert len(self.groups) <= NNZ_MAX_GROUPS and self.ntaps <= NNZ_MAX_TAPS
        d.ntaps_total = self.ntaps
        d.ngroups = len(self.groups)
        d.accumulate = int(self.accumulate)
        ti = 0
        for gi, (ooff, taps) in enumerate(self.groups):
            for a in range(3):
                d.groups[gi].ooff[a] = ooff[a]
            d.groups[gi].tap_begin = ti
            d.groups[gi].ntaps = len(taps)
            for off, widx in taps:
                for a in range(3):
                    d.taps[ti].off[a] = off[a]
                d.taps[ti].widx = widx
                ti += 1
        return d


def _flat(k: Sequence[int], ks: Sequence[int]) -> int:
    return (k[0] * ks[1] + k[1]) * ks[2] + k[2]


def _triple(v) -> Tuple[int, int, int]:
    """int -> (v, v, v); a 2-sequence (2-D layer) -> (1-ish leading axis handled by the caller); 3-sequence as is"""
    if isinstance(v, int):
        return (v, v, v)
    v = tuple(int(i) for i in v)
    assert len(v) == 3, v
    return v


def conv_out_dims(in_dims, ks, stride):
    ks, stride = _triple(ks), _triple(stride)
    return tuple((in_dims[a] + 2 * (ks[a] // 2) - ks[a]) // stride[a] + 1 for a in range(3))


def conv_forward(N, in_dims, Cin, Cout, ks=(3, 3, 3), stride=1, ldi=None, ldo=None, dilation=1) -> TapTable:
    """Y[o] = sum_k X[s*o + (k - ks//2) * dil] W[k] ("same" padding = dil * (ks//2)); packed slice widx == flat kernel
    index.  ks in {1,3}, stride in {1,2}, both per axis (a 2-D layer is the D = 1 case with ks[0] = stride[0] = 1);
    dilation per axis (stride 1 only) - the dilated 3x3 convolutions of REBNCONV, m2net.py:18-30."""
    ks, stride, dil = _triple(ks), _triple(stride), _triple(dilation)
    assert all(d == 1 or st == 1 for d, st in zip(dil, stride)), "dilation is supported for stride-1 axes"
    out_dims = conv_out_dims(in_dims, ks, stride)
    taps = []
    for k in itertools.product(range(ks[0]), range(ks[1]), range(ks[2])):
        off = tuple((k[a] - ks[a] // 2) * dil[a] for a in range(3))
        taps.append((off, _flat(k, ks)))
    return TapTable(N, tuple(in_dims), out_dims, out_dims, Cin, Cout, ldi or Cin, ldo or Cout, stride, (1, 1, 1),
                    [((0, 0, 0), taps)])


def conv_dgrad(N, in_dims, Cin, Cout, ks=(3, 3, 3), stride=1, ldi=None, ldo=None, accumulate=False,
               dilation=1) -> TapTable:
    """dX of the convolution above.  `in` of the table is dY (Cout channels), `out` is dX (Cin channels).

    per axis, stride 1: dX[i] = sum_k dY[i - k + pad] W[k]
    per axis, stride 2: i = 2m + p;  p=0 -> {(k=1, o=m)},  p=1 -> {(k=0, o=m+1), (k=2, o=m)}   (k3, pad 1)
                                       k1: p=0 -> {(k=0, o=m)}, p=1 -> {} (those inputs were never read)
    Output-parity groups are the product over the stride-2 axes (1, 2, 4 or 8 groups).
    ldi / ldo here are the channel strides of dY / dX.
    """
    ks, stride, dil = _triple(ks), _triple(stride), _triple(dilation)
    assert all(d == 1 or st == 1 for d, st in zip(dil, stride)), "dilation is supported for stride-1 axes"
    y_dims = conv_out_dims(in_dims, ks, stride)
    per_axis = []  # parity -> [(off, k)]
    for a in range(3):
        if stride[a] == 1:
            per_axis.append({0: [((ks[a] // 2 - k) * dil[a], k) for k in range(ks[a])]})
        elif stride[a] == 2:
            if ks[a] == 3:
                per_axis.append({0: [(0, 1)], 1: [(1, 0), (0, 2)]})
            elif ks[a] == 1:
                per_axis.append({0: [(0, 0)], 1: []})
            else:
                raise ValueError("kernel size must be 1 or 3")
        else:
            raise ValueError("stride must be 1 or 2")
    groups = []
    for p in itertools.product(*[sorted(pa) for pa in per_axis]):
        taps = []
        for c in itertools.product(per_axis[0][p[0]], per_axis[1][p[1]], per_axis[2][p[2]]):
            off = tuple(c[a][0] for a in range(3))
            k = tuple(c[a][1] for a in range(3))
            taps.append((off, _flat(k, ks)))
        if taps:
            groups.append((p, taps))
    m_dims = tuple((in_dims[a] + stride[a] - 1) // stride[a] for a in range(3))
    return TapTable(N, y_dims, tuple(in_dims), m_dims, Cout, Cin, ldi or Cout, ldo or Cin, (1, 1, 1), stride, groups,
                    accumulate=accumulate)


def dgrad_uncovered(ks, stride) -> bool:
    """True if the data gradient of conv(ks, stride) leaves input positions unwritten (k1 s2 axes: odd positions get
    no contribution) - the caller must zero the destination first."""
    ks, stride = _triple(ks), _triple(stride)
    return any(stride[a] == 2 and ks[a] == 1 for a in range(3))


def conv_wgrad(N, in_dims, Cin, Cout, ks=(3, 3, 3), stride=1, ldx=None, lddy=None, dilation=1) -> TapTable:
    """dW[k][ci][co] = sum_o X[s*o + (k - ks//2) * dil][ci] dY[o][co]: boxed = X, plain = dY."""
    t = conv_forward(N, in_dims, Cin, Cout, ks, stride, ldi=ldx or Cin, ldo=lddy or Cout, dilation=dilation)
    return t


def conv_wgrad_flipped(N, in_dims, Cin, Cout, ks=(3, 3, 3), ldx=None, lddy=None, dilation=1) -> TapTable:
    """The same weight gradient of a STRIDE-1 convolution with the operand roles exchanged:
        dW[k][ci][co] = sum_i X[i][ci] dY[i - (k - ks//2) * dil][co]        boxed = dY (A = Cout), plain = X (B = Cin)
    (substitute i = o + off; out-of-range dY positions contribute nothing, exactly like the zero padding of X in the
    direct form).  The result block is dW[widx][co][ci].  Used when X is a RAW conv output that the kernel normalises while
    staging (round 4): the plain operand has no halo, so every voxel of X is normalised once per tile instead of 2.3 times."""
    ks = _triple(ks)
    t = conv_forward(N, in_dims, Cout, Cin, ks, 1, ldi=lddy or Cout, ldo=ldx or Cin, dilation=dilation)
    (ooff, taps), = t.groups
    t.groups = [(ooff, [(tuple(-o for o in off), widx) for off, widx in taps])]
    return t


def convT_forward(N, in_dims, Cin, Cout, ldi=None, ldo=None, stride=2) -> TapTable:
    """ConvTranspose(kernel = stride, per axis 1 or 2): out[s*m + p] = bias + sum_ci in[m] W[ci][co][p];
    prod(stride) one-tap groups."""
    st = _triple(stride)
    out_dims = tuple(st[a] * in_dims[a] for a in range(3))
    groups = []
    for p in itertools.product(range(st[0]), range(st[1]), range(st[2])):
        groups.append((p, [((0, 0, 0), _flat(p, st))]))
    return TapTable(N, tuple(in_dims), out_dims, tuple(in_dims), Cin, Cout, ldi or Cin, ldo or Cout, (1, 1, 1), st,
                    groups)


def convT_dgrad(N, in_dims, Cin, Cout, ldi=None, ldo=None, accumulate=False, stride=2) -> TapTable:
    """dIn[m][ci] = sum_p sum_co dOut[s*m + p][co] W[ci][co][p]; `in` = dOut (Cout ch), `out` = dIn (Cin ch)."""
    st = _triple(stride)
    out_dims = tuple(st[a] * in_dims[a] for a in range(3))
    taps = [(p, _flat(p, st)) for p in itertools.product(range(st[0]), range(st[1]), range(st[2]))]
    return TapTable(N, out_dims, tuple(in_dims), tuple(in_dims), Cout, Cin, ldi or Cout, ldo or Cin, st, (1, 1, 1),
                    [((0, 0, 0), taps)], accumulate=accumulate)


def convT_wgrad(N, in_dims, Cin, Cout, lddout=None, ldin=None, stride=2) -> TapTable:
    """dW[ci][co][p] = sum_m in[m][ci] dOut[s*m + p][co]: boxed = dOut (A = Cout), plain = in (B = Cin)."""
    return convT_dgrad(N, in_dims, Cin, Cout, ldi=lddout or Cout, ldo=ldin or Cin, stride=stride)


def ksel_array(ksel: Sequence[int]):
    arr = (C.c_int * 32)()
    for i, k in enumerate(ksel):
        arr[i] = k
    return arr
