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
from dataclasses import dataclass, field
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
    in_stride: int
    out_stride: int
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
        d.Cin, d.Cout, d.ldi, d.ldo = self.Cin, self.Cout, self.ldi, self.ldo
        d.in_stride, d.out_stride = self.in_stride, self.out_stride
        offs = [t[0] for g in self.groups for t in g[1]]
        lo = [min(o[a] for o in offs) for a in range(3)]
        hi = [max(o[a] for o in offs) for a in range(3)]
        d.ext = max(hi[a] - lo[a] for a in range(3))
        for a in range(3):
            d.lo[a] = lo[a]
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


def conv_out_dims(in_dims, ks, stride):
    return tuple((in_dims[a] + 2 * (ks[a] // 2) - ks[a]) // stride + 1 for a in range(3))


def conv_forward(N, in_dims, Cin, Cout, ks=(3, 3, 3), stride=1, ldi=None, ldo=None) -> TapTable:
    """Y[o] = sum_k X[s*o + k - pad] W[k]; packed slice widx == flat kernel index."""
    out_dims = conv_out_dims(in_dims, ks, stride)
    taps = []
    for k in itertools.product(range(ks[0]), range(ks[1]), range(ks[2])):
        off = tuple(k[a] - ks[a] // 2 for a in range(3))
        taps.append((off, _flat(k, ks)))
    nk = ks[0] * ks[1] * ks[2]
    return TapTable(N, tuple(in_dims), out_dims, out_dims, Cin, Cout, ldi or Cin, ldo or Cout, stride, 1,
                    [((0, 0, 0), taps)])


def conv_dgrad(N, in_dims, Cin, Cout, ks=(3, 3, 3), stride=1, ldi=None, ldo=None, accumulate=False) -> TapTable:
    """dX of the convolution above.  `in` of the table is dY (Cout channels), `out` is dX (Cin channels).

    stride 1: dX[i] = sum_k dY[i - k + pad] W[k]
    stride 2: i = 2m + p;  per axis p=0 -> {(k=1, o=m)},  p=1 -> {(k=0, o=m+1), (k=2, o=m)}   (k3, pad 1)
    ldi / ldo here are the channel strides of dY / dX.
    """
    y_dims = conv_out_dims(in_dims, ks, stride)
    nk = ks[0] * ks[1] * ks[2]
    if stride == 1:
        taps = []
        for k in itertools.product(range(ks[0]), range(ks[1]), range(ks[2])):
            off = tuple(ks[a] // 2 - k[a] for a in range(3))
            taps.append((off, _flat(k, ks)))
        groups = [((0, 0, 0), taps)]
        m_dims = tuple(in_dims)
        os_ = 1
    else:
        assert stride == 2
        per_axis = []
        for a in range(3):
            if ks[a] == 3:
                per_axis.append({0: [(0, 1)], 1: [(1, 0), (0, 2)]})  # parity -> [(off, k)]
            elif ks[a] == 1:
                per_axis.append({0: [(0, 0)], 1: []})
            else:
                raise ValueError("kernel size must be 1 or 3")
        groups = []
        for p in itertools.product((0, 1), (0, 1), (0, 1)):
            taps = []
            for c in itertools.product(per_axis[0][p[0]], per_axis[1][p[1]], per_axis[2][p[2]]):
                off = tuple(c[a][0] for a in range(3))
                k = tuple(c[a][1] for a in range(3))
                taps.append((off, _flat(k, ks)))
            if taps:
                groups.append((p, taps))
        m_dims = tuple((in_dims[a] + 1) // 2 for a in range(3))
        os_ = 2
    return TapTable(N, y_dims, tuple(in_dims), m_dims, Cout, Cin, ldi or Cout, ldo or Cin, 1, os_, groups,
                    accumulate=accumulate)


def conv_wgrad(N, in_dims, Cin, Cout, ks=(3, 3, 3), stride=1, ldx=None, lddy=None) -> TapTable:
    """dW[k][ci][co] = sum_o X[s*o + k - pad][ci] dY[o][co]: boxed = X, plain = dY."""
    t = conv_forward(N, in_dims, Cin, Cout, ks, stride, ldi=ldx or Cin, ldo=lddy or Cout)
    return t


def convT_forward(N, in_dims, Cin, Cout, ldi=None, ldo=None) -> TapTable:
    """ConvTranspose3d(k=2, s=2): out[2m + p] = bias + sum_ci in[m] W[ci][co][p]; 8 one-tap groups."""
    out_dims = tuple(2 * d for d in in_dims)
    groups = []
    for p in itertools.product((0, 1), (0, 1), (0, 1)):
        groups.append((p, [((0, 0, 0), _flat(p, (2, 2, 2)))]))
    return TapTable(N, tuple(in_dims), out_dims, tuple(in_dims), Cin, Cout, ldi or Cin, ldo or Cout, 1, 2, groups)


def convT_dgrad(N, in_dims, Cin, Cout, ldi=None, ldo=None, accumulate=False) -> TapTable:
    """dIn[m][ci] = sum_p sum_co dOut[2m + p][co] W[ci][co][p]; `in` = dOut (Cout ch), `out` = dIn (Cin ch)."""
    out_dims = tuple(2 * d for d in in_dims)
    taps = [(p, _flat(p, (2, 2, 2))) for p in itertools.product((0, 1), (0, 1), (0, 1))]
    return TapTable(N, out_dims, tuple(in_dims), tuple(in_dims), Cout, Cin, ldi or Cout, ldo or Cin, 2, 1,
                    [((0, 0, 0), taps)], accumulate=accumulate)


def convT_wgrad(N, in_dims, Cin, Cout, lddout=None, ldin=None) -> TapTable:
    """dW[ci][co][p] = sum_m in[m][ci] dOut[2m + p][co]: boxed = dOut (A = Cout), plain = in (B = Cin)."""
    return convT_dgrad(N, in_dims, Cin, Cout, ldi=lddout or Cout, ldo=ldin or Cin)


def ksel_array(ksel: Sequence[int]):
    arr = (C.c_int * 32)()
    for i, k in enumerate(ksel):
        arr[i] = k
    return arr
