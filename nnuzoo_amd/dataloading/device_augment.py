"""Device-side training augmentations: the interpolating and intensity transforms of the reference's
training chain - `nnUNetTrainer.get_training_transforms`, /root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainer.py:825-973 -
on a batch that `nnUNetDataLoaderDevice` has just cut in HBM, as HIP launches (csrc/augment.hip), no worker processes.

Order, probabilities and parameter ranges are the call site's (line numbers of nnUNetTrainer.py):
    SpatialTransform   :845-852  p_rotation 0.2 (angles from `rotation_for_DA`), p_scaling 0.2 in (0.7, 1.4), axes synchronised,
                                 no elastic deformation, data bi/trilinear with zeros outside, segmentation nearest with -1 outside
    GaussianNoise      :857-863  p 0.1, variance U(0, 0.1), one draw for all channels
    GaussianBlur       :864-871  p 0.2, then per channel p 0.5: sigma per channel AND axis from (0.5, 1), separable, edges repeated
    MultiplicativeBrightness :872-878  p 0.15, multiplier per channel from (0.75, 1.25)
    Contrast           :879-886  p 0.15, factor per channel from (0.75, 1.25), range preserved
    SimulateLowResolution :887-896  p 0.25, then per channel p 0.5: scale from (0.5, 1) for all axes, nearest down / linear up
                                 (the package goes up with a cubic)
    Gamma (inverted)   :897-905  p 0.1, gamma per channel from (0.7, 1.5), mean / std retained
    Gamma              :906-914  p 0.3, the same without the inversion
    (MirrorTransform :915-920 is folded into the loader's crop; DownsampleSegForDS :971 follows in the loader)
    RemoveLabelTansform(-1, 0) :929-931
Every `RandomTransform(..., apply_probability=p)` is decided per SAMPLE (the reference's workers push one sample at a time through
the chain).  "(lo, hi)" ranges marked BGContrast at the call site draw from [lo, 1) with probability 1/2 and from [1, hi) otherwise.

The transform classes are batchgeneratorsv2's (`pyproject.toml:51`), absent from /root/reference and from this image: what each
one computes is restated here from its published algorithm and the call site - PARITY UNPINNED; `tests/test_device_augment_gpu.py`
holds every launch to a plain torch fp32 formulation of the same arithmetic.  Two draws were corrected in round 6 after review
(both from the reviewer's and the author's recollection of the package, neither can be diffed here): Gaussian noise takes
the sampled `noise_variance` as the standard deviation (sigma = U(0, 0.1), not its square root), and the spatial transform's
`scaling=(0.7, 1.4)` is a plain tuple drawn uniformly (the below / above 1 split is BGContrast's, used only where the call site wraps
the range in BGContrast).  Host draws use a private numpy RandomState
(reproducible from `seed`; not the reference's torch / numpy call sequence)."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _lib
from .._lib import call, ptr, stream_ptr

OP_NOISE, OP_LINEAR, OP_CONTRAST, OP_GAMMA, OP_RESTORE = 0, 1, 2, 3, 4


def _rot3(ax: int, t: float) -> np.ndarray:
    c, s = np.cos(t), np.sin(t)
    m = np.eye(3)
    i, j = [(1, 2), (0, 2), (0, 1)][ax]          # rotation in the plane of the two OTHER axes of (z, y, x)
    m[i, i], m[i, j], m[j, i], m[j, j] = c, -s, s, c
    return m


class DeviceAugmenter:
    """`data, seg = augmenter(data, seg)`: data float32 (B, C, *patch) CUDA, seg int16 (B, 1, *patch) CUDA or None; both
    contiguous; returns tensors of the same shapes (data is transformed in place where no resampling is drawn)."""

    p_rotation, p_scaling, scaling = 0.2, 0.2, (0.7, 1.4)
    p_noise, noise_variance = 0.1, (0.0, 0.1)
    p_blur, p_blur_per_channel, blur_sigma = 0.2, 0.5, (0.5, 1.0)
    p_brightness, brightness = 0.15, (0.75, 1.25)
    p_contrast, contrast = 0.15, (0.75, 1.25)
    p_lowres, p_lowres_per_channel, lowres_scale = 0.25, 0.5, (0.5, 1.0)
    p_gamma_inverted, p_gamma, gamma = 0.1, 0.3, (0.7, 1.5)

    def __init__(self, patch_size: Sequence[int], rotation_for_DA: Tuple[float, float], do_dummy_2d_data_aug: bool = False,
                 seed: Optional[int] = None, use_mask_for_norm: Optional[Sequence[bool]] = None,
                 cascade_labels: Optional[Sequence[int]] = None, regions: Optional[Sequence] = None,
                 ignore_label: Optional[int] = None):
        """use_mask_for_norm / cascade_labels (`foreground_labels` of a cascaded configuration) / regions / ignore_label: the arguments
        of nnUNetTrainer.get_training_transforms (:825-835) that switch on MaskTransform (:921-927), MoveSegAsOneHotToData (:932-939)
        and ConvertSegmentationToRegionsTransform (:961-969).  The two random morphology transforms of the cascade (:940-959:
        skimage / connected components on the CPU) are not built; the one-hot move itself is."""
        self.use_mask_for_norm = list(use_mask_for_norm) if use_mask_for_norm is not None and any(use_mask_for_norm) else None
        self.cascade_labels = [int(v) for v in cascade_labels] if cascade_labels else None
        self.regions = None
        if regions is not None:
            regs = list(regions) + ([ignore_label] if ignore_label is not None else [])
            self.regions = [tuple(int(v) for v in (r if isinstance(r, (list, tuple)) else (r,))) for r in regs]
        self.patch_size = tuple(int(i) for i in patch_size)
        self.rotation = (float(rotation_for_DA[0]), float(rotation_for_DA[1]))
        self.dummy_2d = bool(do_dummy_2d_data_aug)
        self.rng = np.random.RandomState(seed)
        self.last = {}                      # what the most recent call drew (tests, logging)
        self._ws = None
        self._calls = 0

    # ---- draws ------------------------------------------------------------------------------------------------------------
    def _bg_range(self, r):
        lo, hi = r
        if self.rng.uniform() < 0.5 and lo < 1:
            return self.rng.uniform(lo, 1.0)
        return self.rng.uniform(max(lo, 1.0), hi)

    def _draw_matrix(self, nd: int):
        """index-space map of one sample: source = M (out - centre) + centre; None = identity (nothing drawn)"""
        m = np.eye(3)
        drawn = False
        if self.rng.uniform() < self.p_rotation:
            drawn = True
            if nd == 2 or self.dummy_2d:
                m = _rot3(0, self.rng.uniform(*self.rotation)) @ m          # in-plane (y, x) only
            else:
                for ax in range(3):
                    m = _rot3(ax, self.rng.uniform(*self.rotation)) @ m
        if self.rng.uniform() < self.p_scaling:
            drawn = True
            # the call site passes a plain tuple (nnUNetTrainer.py:856 `scaling=(0.7, 1.4)`), which batchgeneratorsv2's
            # sample_scalar draws UNIFORMLY over the interval; the 50/50 split below / above 1 is BGContrast's rule and applies
            # only to the ranges the call site wraps in BGContrast (brightness, contrast, gamma: _bg_range) - ADVICE r5
            lo, hi = self.scaling
            sc = self.rng.uniform(lo, hi)
            s = np.diag([1.0 if (nd == 2 or self.dummy_2d) else sc, sc, sc])
            m = m @ s
        return m if drawn else None

    # ---- launches ---------------------------------------------------------------------------------------------------------
    def _stats(self, x: torch.Tensor, nbc: int, n: int) -> torch.Tensor:
        lib = _lib.load()
        need = int(lib.nnz_aug_stats_workspace_floats(nbc))
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = torch.empty(need, dtype=torch.float32, device=x.device)
        st = torch.empty((nbc, 4), dtype=torch.float32, device=x.device)
        call("nnz_aug_stats_f32", ptr(x), n, nbc, ptr(self._ws), ptr(st), stream_ptr())
        return st

    def _op(self, x, nbc, n, op, rec: np.ndarray, sa=None, sb=None, seed=0):
        r = torch.tensor(rec, dtype=torch.float32).to(x.device)        # (a copy: `rec` is rewritten for the next transform)
        call("nnz_aug_intensity_f32", ptr(x), n, nbc, op, ptr(r), ptr(sa), ptr(sb), int(seed) & 0x7fffffff, stream_ptr())

    def __call__(self, data: torch.Tensor, seg: Optional[torch.Tensor]):
        if not data.is_cuda or data.dtype != torch.float32 or not data.is_contiguous():
            raise RuntimeError("DeviceAugmenter: contiguous float32 CUDA data expected (there is no CPU path)")
        if seg is not None and (not seg.is_cuda or seg.dtype != torch.int16 or not seg.is_contiguous()):
            raise RuntimeError("DeviceAugmenter: contiguous int16 CUDA segmentation expected")
        B, C = data.shape[:2]
        sp = tuple(data.shape[2:])
        nd = len(sp)
        D, H, W = ((1,) + sp) if nd == 2 else sp
        n, nbc = D * H * W, B * C
        self._calls += 1
        last = {"matrices": [None] * B}
        # -- SpatialTransform
        mats = [self._draw_matrix(nd) for _ in range(B)]
        last["matrices"] = mats
        if any(m is not None for m in mats):
            flat = np.zeros((B, 12), dtype=np.float32)
            for b, m in enumerate(mats):
                mm = np.eye(3) if m is None else m
                flat[b].reshape(3, 4)[:, :3] = mm
            out = torch.empty_like(data)
            call("nnz_aug_affine_f32", ptr(data), ptr(out), flat.ctypes.data, B, C, D, H, W, 0.0, stream_ptr())
            data = out
            if seg is not None:
                so = torch.empty_like(seg)
                call("nnz_aug_affine_i16", ptr(seg), ptr(so), flat.ctypes.data, B, seg.shape[1], D, H, W, -1, stream_ptr())
                seg = so
        rec = np.zeros((nbc, 4), dtype=np.float32)

        def per_sample(p):
            return [self.rng.uniform() < p for _ in range(B)]
        # -- GaussianNoise (one variance per sample, all channels)
        on = per_sample(self.p_noise)
        last["noise_sigma"] = [None] * B
        if any(on):
            rec[:] = 0
            for b in range(B):
                if on[b]:
                    # batchgeneratorsv2's GaussianNoiseTransform hands the sampled `noise_variance` to torch.normal AS THE
                    # STANDARD DEVIATION (sigma = U(0, 0.1)); batchgenerators v1 took its square root - ADVICE r5
                    sig = float(self.rng.uniform(*self.noise_variance))
                    last["noise_sigma"][b] = sig
                    rec[b * C:(b + 1) * C, 0], rec[b * C:(b + 1) * C, 1] = 1.0, sig
            last["noise_seed"] = int(self.rng.randint(0, 2 ** 31 - 1))
            self._op(data, nbc, n, OP_NOISE, rec, seed=last["noise_seed"])
        # -- GaussianBlur (per channel with probability 0.5; sigma per channel and axis)
        on = per_sample(self.p_blur)
        last["blur_sigma"] = np.full((B, C, 3), np.nan)
        if any(on):
            rec[:] = 0
            for b in range(B):
                if on[b]:
                    for c in range(C):
                        if self.rng.uniform() < self.p_blur_per_channel:
                            sg = [self.rng.uniform(*self.blur_sigma) for _ in range(3)]
                            if nd == 2 or self.dummy_2d:
                                sg[0] = 0.0
                            last["blur_sigma"][b, c] = sg
                            rec[b * C + c] = (1.0, *sg)
            if rec[:, 0].any():
                r = torch.tensor(rec, dtype=torch.float32).to(data.device)
                tmp = torch.empty_like(data)
                for axis in range(3):
                    if (D, H, W)[axis] == 1:
                        continue
                    call("nnz_aug_blur_axis_f32", ptr(data), ptr(tmp), nbc, D, H, W, axis, ptr(r), stream_ptr())
                    data, tmp = tmp, data
        # -- MultiplicativeBrightness (per channel)
        on = per_sample(self.p_brightness)
        last["brightness"] = np.full((B, C), np.nan)
        if any(on):
            rec[:] = 0
            for b in range(B):
                if on[b]:
                    for c in range(C):
                        mlt = self._bg_range(self.brightness)
                        last["brightness"][b, c] = mlt
                        rec[b * C + c, :3] = (1.0, mlt, 0.0)
            self._op(data, nbc, n, OP_LINEAR, rec)
        # -- Contrast (per channel, range preserved)
        on = per_sample(self.p_contrast)
        last["contrast"] = np.full((B, C), np.nan)
        if any(on):
            rec[:] = 0
            for b in range(B):
                if on[b]:
                    for c in range(C):
                        f = self._bg_range(self.contrast)
                        last["contrast"][b, c] = f
                        rec[b * C + c, :2] = (1.0, f)
            self._op(data, nbc, n, OP_CONTRAST, rec, sa=self._stats(data, nbc, n))
        # -- SimulateLowResolution (per channel with probability 0.5; one scale for all axes)
        on = per_sample(self.p_lowres)
        last["lowres_scale"] = np.full((B, C), np.nan)
        if any(on):
            rec[:] = 0
            for b in range(B):
                if on[b]:
                    for c in range(C):
                        if self.rng.uniform() < self.p_lowres_per_channel:
                            sc = self.rng.uniform(*self.lowres_scale)
                            last["lowres_scale"][b, c] = sc
                            rec[b * C + c, :2] = (1.0, sc)
            if rec[:, 0].any():
                r = torch.tensor(rec, dtype=torch.float32).to(data.device)
                out = torch.empty_like(data)
                call("nnz_aug_lowres_f32", ptr(data), ptr(out), nbc, D, H, W, int(nd == 2 or self.dummy_2d), ptr(r), stream_ptr())
                data = out
        # -- Gamma, inverted image first (:897-905), then plain (:906-914); mean / std retained
        for name, p, invert in (("gamma_inverted", self.p_gamma_inverted, True), ("gamma", self.p_gamma, False)):
            on = per_sample(p)
            last[name] = np.full((B, C), np.nan)
            if not any(on):
                continue
            rec[:] = 0
            for b in range(B):
                if on[b]:
                    for c in range(C):
                        g = self._bg_range(self.gamma)
                        last[name][b, c] = g
                        rec[b * C + c, :2] = (1.0, g)
            neg = rec.copy()
            neg[:, 1], neg[:, 2] = -1.0, 0.0
            if invert:
                self._op(data, nbc, n, OP_LINEAR, neg)
            before = self._stats(data, nbc, n)
            self._op(data, nbc, n, OP_GAMMA, rec, sa=before)
            self._op(data, nbc, n, OP_RESTORE, rec, sa=self._stats(data, nbc, n), sb=before)
            if invert:
                self._op(data, nbc, n, OP_LINEAR, neg)
        # -- MaskTransform (:921-927): normalisation masks - everything outside the mask (seg < 0) back to 0 in the masked channels
        if self.use_mask_for_norm is not None and seg is not None:
            data = mask_outside(data, seg, [i for i, m in enumerate(self.use_mask_for_norm) if m], 0, 0.0)
        # -- RemoveLabelTansform(-1, 0)
        if seg is not None:
            call("nnz_aug_relabel_i16", ptr(seg), seg.numel(), -1, 0, stream_ptr())
        # -- cascade: the previous stage's segmentation (seg channel 1) as one-hot image channels (:932-939)
        if self.cascade_labels is not None and seg is not None:
            data, seg = move_seg_as_onehot_to_data(data, seg, 1, self.cascade_labels, remove_from_origin=True)
        # -- region-based training (:961-969): label map -> one binary map per region (the ignore label as the last region)
        if self.regions is not None and seg is not None:
            seg = seg_to_regions(seg, self.regions, 0)
        self.last = last
        return data, seg


# ---- label-side transforms the reference defines itself (custom_transforms/*.py), as launches on the resident batch -------------
def _flat(t: torch.Tensor):
    return t.shape[0], t.shape[1], int(np.prod(t.shape[2:]))


def seg_to_regions(seg: torch.Tensor, regions, seg_channel: int = 0) -> torch.Tensor:
    """ConvertSegmentationToRegionsTransform (region_based_training.py:7-39): (B, Cs, *sp) int16 -> (B, R, *sp) int16 of 0 / 1"""
    import ctypes as C
    if not seg.is_cuda or seg.dtype != torch.int16 or not seg.is_contiguous():
        raise RuntimeError("seg_to_regions: contiguous int16 CUDA segmentation expected (there is no CPU path)")
    B, Cs, n = _flat(seg)
    regs = [tuple(r) if isinstance(r, (list, tuple)) else (r,) for r in regions]
    begin = np.zeros(len(regs) + 1, dtype=np.int32)
    begin[1:] = np.cumsum([len(r) for r in regs])
    labels = np.array([v for r in regs for v in r], dtype=np.int32)
    out = torch.empty((B, len(regs)) + tuple(seg.shape[2:]), dtype=torch.int16, device=seg.device)
    call("nnz_aug_seg_to_regions_i16", ptr(seg), ptr(out), B, Cs, int(seg_channel), n, begin.ctypes.data, labels.ctypes.data, len(regs),
         stream_ptr())
    return out


def move_seg_as_onehot_to_data(data: torch.Tensor, seg: torch.Tensor, index_in_origin: int, all_labels,
                               remove_from_origin: bool = True):
    """MoveSegAsOneHotToData (cascade_transforms.py:10-39): one-hot of seg[:, index_in_origin] over `all_labels` appended to data"""
    if not (data.is_cuda and seg.is_cuda and data.dtype == torch.float32 and seg.dtype == torch.int16 and data.is_contiguous()
            and seg.is_contiguous()):
        raise RuntimeError("move_seg_as_onehot_to_data: contiguous float32 data / int16 seg on the device expected")
    B, Cd, n = _flat(data)
    _, Cs, _ = _flat(seg)
    labels = np.array([int(v) for v in all_labels], dtype=np.int32)
    wide = torch.empty((B, Cd + len(labels)) + tuple(data.shape[2:]), dtype=torch.float32, device=data.device)
    wide[:, :Cd].copy_(data)
    call("nnz_aug_seg_onehot_to_data_f32", ptr(seg), ptr(wide), B, Cs, int(index_in_origin), Cd + len(labels), Cd, n,
         labels.ctypes.data, len(labels), stream_ptr())
    if remove_from_origin:
        seg = seg[:, [i for i in range(Cs) if i != index_in_origin]].contiguous()
    return wide, seg


def mask_outside(data: torch.Tensor, seg: torch.Tensor, apply_to_channels, mask_idx_in_seg: int = 0, set_outside_to: float = 0.0):
    """MaskTransform (masking.py:6-24): data[:, c][seg[:, mask_idx_in_seg] < 0] = set_outside_to, in place"""
    if not (data.is_cuda and seg.is_cuda and data.dtype == torch.float32 and seg.dtype == torch.int16 and data.is_contiguous()
            and seg.is_contiguous()):
        raise RuntimeError("mask_outside: contiguous float32 data / int16 seg on the device expected")
    B, Cd, n = _flat(data)
    bits = 0
    for c in apply_to_channels:
        bits |= 1 << int(c)
    call("nnz_aug_mask_outside_f32", ptr(data), ptr(seg), B, Cd, seg.shape[1], int(mask_idx_in_seg), n, bits, float(set_outside_to),
         stream_ptr())
    return data
