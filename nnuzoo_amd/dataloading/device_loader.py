"""Device-resident training data loader - the MI355X counterpart of
`nnUNetDataLoader` (/root/reference/nnunetv2/training/dataloading/data_loader.py:19-262).

The reference cuts patches on CPU worker processes (numpy crop_and_pad_nd per sample, torch transforms with one thread,
batchgenerators' multi-threaded augmenter around it) and ships every batch over PCIe.  At 140 patches/s per GPU x 8 GPUs
those workers, not the GPUs, set the pace.  Here the preprocessed cases are uploaded ONCE (288 GB of HBM holds whole
datasets - `DeviceCaseStore`), a step's host work is drawing B bounding boxes with the reference's own rule and RNG call
order (`get_bbox`, data_loader.py:102-178, restated line by line: equal numpy seeds give equal boxes -
tests/golden/dataloader_bbox.json comes from the reference's function), and the voxels move in two HIP launches
(`nnz_crop_pad_f32 / _i16`, csrc/input_pipeline.hip) plus one per deep-supervision scale (`nnz_downsample_nearest_i16`).

What is covered of the reference's transform chain (nnUNetTrainer.get_training_transforms, nnUNetTrainer.py:860-973): the
voxel-moving transforms that need no interpolation - MirrorTransform (folded into the crop's index arithmetic) and
DownsampleSegForDSTransform - and, since round 5, through `augmenter=DeviceAugmenter(...)` (dataloading/device_augment.py,
csrc/augment.hip): SpatialTransform (rotation / scaling), Gaussian noise, Gaussian blur, multiplicative brightness, contrast, low-resolution
simulation, both gamma transforms and RemoveLabel(-1 -> 0) with the call site's probabilities and ranges (batchgeneratorsv2 is absent from the reference
tree and from this image: arithmetic restated, PARITY UNPINNED), Gaussian blur and low-resolution simulation included.
Anything else can be appended as `transforms`, a callable on the device batch.

Same constructor arguments and batch contract as the reference class: `{'data': float32 (B, C, *patch), 'target': int16
tensor or list of tensors per deep-supervision scale, 'keys'}` - but the tensors are CUDA tensors, ready for train_step.
"""
from __future__ import annotations

import ctypes as C
import warnings
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .._lib import call, ptr, stream_ptr


class DeviceCaseStore:
    """cases resident in HBM.  `dataset`: anything with `.identifiers` and `.load_case(id) -> (data, seg, seg_prev,
    properties)` (the reference's nnUNetBaseDataset contract, nnunet_dataset.py:21-60): data float (C, *shape), seg integer
    (1, *shape) or None, properties with 'class_locations'."""

    def __init__(self, dataset, device: Union[str, torch.device] = "cuda"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceCaseStore keeps the cases in HBM; there is no CPU path")
        self.identifiers = list(dataset.identifiers)
        self.data, self.seg, self.properties = {}, {}, {}
        for k in self.identifiers:
            data, seg, seg_prev, props = dataset.load_case(k)
            if seg_prev is not None:
                raise NotImplementedError("cascade (previous-stage segmentations) is outside the hot path")
            d = torch.as_tensor(np.ascontiguousarray(data), dtype=torch.float32)
            if d.dim() == 3:                       # 2-D cases are kept as depth-1 volumes
                d = d[:, None]
            self.data[k] = d.to(self.device)
            if seg is not None:
                s = torch.as_tensor(np.ascontiguousarray(seg).astype(np.int16))
                if s.dim() == 3:
                    s = s[:, None]
                self.seg[k] = s.to(self.device)
            self.properties[k] = props

    def load_case(self, k):
        return self.data[k], self.seg.get(k), None, self.properties[k]

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in list(self.data.values()) + list(self.seg.values()))


def _bbox_bounds(data_shape, patch_size, need_to_pad):
    """Inclusive range of admissible lower patch corners per axis.  A case smaller than the patch is padded on both sides
    (the extra voxel of an odd amount goes to the upper side); `need_to_pad` widens the range symmetrically."""
    shape = np.asarray(data_shape, dtype=np.int64)
    patch = np.asarray(patch_size, dtype=np.int64)
    pad = np.maximum(np.asarray(need_to_pad, dtype=np.int64), patch - shape)
    lo = np.floor_divide(-pad, 2)
    hi = shape + pad // 2 + pad % 2 - patch
    return [int(v) for v in lo], [int(v) for v in hi]


def _class_to_centre_on(annotated_key, force_fg: bool, class_locations, overwrite_class, verbose: bool):
    """Which entry of `class_locations` the patch centre is drawn from; None = plain random crop.  Draws from the numpy RNG
    only when a foreground class has to be chosen among several candidates."""
    if not force_fg:
        # only reached with an ignore label: centre on any annotated voxel so that the patch is not all-ignore
        if len(class_locations[annotated_key]) == 0:
            warnings.warn('Warning! No annotated pixels in image!')
            return None
        return annotated_key
    assert class_locations is not None, 'if force_fg is set class_locations cannot be None'
    if overwrite_class is not None:
        assert overwrite_class in class_locations.keys(), \
            'desired class ("overwrite_class") does not have class_locations (missing key)'
    candidates = [k for k, locs in class_locations.items() if len(locs) > 0]

    # the "all annotated voxels" region is a candidate only when nothing else is (keys may be ints or tuples)
    def is_region_key(k):
        return isinstance(k, tuple) and k == annotated_key
    if len(candidates) > 1 and any(is_region_key(k) for k in candidates):
        candidates = [k for k in candidates if not is_region_key(k)]
    if not candidates:
        if verbose:
            print('case does not contain any foreground classes')
        return None
    if overwrite_class is not None and overwrite_class in candidates:
        return overwrite_class
    return candidates[np.random.choice(len(candidates))]


class nnUNetDataLoader:
    def __init__(self, data: DeviceCaseStore, batch_size: int, initial_patch_size, final_patch_size, label_manager,
                 oversample_foreground_percent: float = 0.0, sampling_probabilities=None, pad_sides=None,
                 probabilistic_oversampling: bool = False, transforms=None, target_type: str = "segmentation",
                 deep_supervision_scales: Optional[Sequence[Sequence[float]]] = None,
                 mirror_axes: Optional[Tuple[int, ...]] = None, augmenter=None):
        if target_type != "segmentation":
            raise NotImplementedError("device loader: segmentation targets only (the training hot path)")
        if not isinstance(data, DeviceCaseStore):
            data = DeviceCaseStore(data)
        self._data, self.batch_size = data, batch_size
        # 2-D patch sizes become pseudo 3-D, the singleton axis is removed before returning (data_loader.py:39-45)
        if len(initial_patch_size) == 2:
            final_patch_size, initial_patch_size = (1, *final_patch_size), (1, *initial_patch_size)
            self.patch_size_was_2d = True
        else:
            self.patch_size_was_2d = False
        self.indices = data.identifiers
        self.oversample_foreground_percent = oversample_foreground_percent
        self.final_patch_size = self.patch_size = tuple(int(i) for i in final_patch_size)
        self.initial_patch_size = tuple(int(i) for i in initial_patch_size)
        self.need_to_pad = (np.array(initial_patch_size) - np.array(final_patch_size)).astype(int)     # :54-59
        if pad_sides is not None:
            if self.patch_size_was_2d:
                pad_sides = (0, *pad_sides)
            for d in range(len(self.need_to_pad)):
                self.need_to_pad[d] += pad_sides[d]
        self.pad_sides = pad_sides
        self.sampling_probabilities = sampling_probabilities
        if label_manager is not None:
            self.annotated_classes_key = tuple([-1] + list(label_manager.all_labels))
            self.has_ignore = label_manager.has_ignore_label
        else:
            self.annotated_classes_key, self.has_ignore = tuple(), False
        self.get_do_oversample = self._probabilistic_oversampling if probabilistic_oversampling \
            else self._oversample_last_XX_percent
        self.transforms = transforms
        # `augmenter(data, seg) -> (data, seg)` on the full-resolution device batch, BEFORE the deep-supervision targets are cut
        # (dataloading/device_augment.DeviceAugmenter: the reference chain's spatial / intensity transforms as HIP launches)
        self.augmenter = augmenter
        self.deep_supervision_scales = deep_supervision_scales
        self.mirror_axes = tuple(mirror_axes) if mirror_axes else None
        first = data.data[self.indices[0]]
        self.num_channels = first.shape[0]
        self.data_shape = (batch_size, self.num_channels, *self.patch_size)

    # ---- host logic restated from the reference ------------------------------------------------------------------------
    def _oversample_last_XX_percent(self, sample_idx: int) -> bool:                                     # :77-81
        return not sample_idx < round(self.batch_size * (1 - self.oversample_foreground_percent))

    def _probabilistic_oversampling(self, sample_idx: int) -> bool:                                     # :83-85
        return np.random.uniform() < self.oversample_foreground_percent

    def get_indices(self):
        """batchgenerators' DataLoader.get_indices with infinite=True (the reference's ctor call, :35-36; the package is
        absent here: restated, unpinned)"""
        return np.random.choice(self.indices, self.batch_size, replace=True, p=self.sampling_probabilities)

    def get_bbox(self, data_shape, force_fg: bool, class_locations, overwrite_class=None, verbose: bool = False):
        """Patch corner rule of data_loader.py:102-178 (bounds, class selection, voxel draw).  The order of the numpy RNG
        draws is part of the contract - class choice, then voxel choice, or one randint per axis - and is pinned against the
        reference's own method by tests/golden/dataloader_bbox.json."""
        lo, hi = _bbox_bounds(data_shape, self.patch_size, self.need_to_pad)
        axes = range(len(data_shape))
        centre_class = None
        if force_fg or self.has_ignore:
            centre_class = _class_to_centre_on(self.annotated_classes_key, force_fg, class_locations, overwrite_class, verbose)
        if centre_class is None:
            corner = [np.random.randint(lo[a], hi[a] + 1) for a in axes]
        else:
            locs = class_locations[centre_class]
            voxel = locs[np.random.choice(len(locs))]          # (channel, *spatial): spatial index a sits at a + 1
            corner = [max(lo[a], voxel[a + 1] - self.patch_size[a] // 2) for a in axes]
        return corner, [corner[a] + self.patch_size[a] for a in axes]

    # ---- the batch ----------------------------------------------------------------------------------------------------------
    def _draw_flips(self) -> List[int]:
        """MirrorTransform: every allowed axis is flipped with probability 0.5, per sample (batchgeneratorsv2 - absent,
        unpinned; axes are spatial axes of the PATCH as the network sees it)"""
        if not self.mirror_axes:
            return [0] * self.batch_size
        off = 1 if self.patch_size_was_2d else 0
        out = []
        for _ in range(self.batch_size):
            m = 0
            for ax in self.mirror_axes:
                if np.random.uniform() < 0.5:
                    m |= 1 << (ax + off)
            out.append(m)
        return out

    def generate_train_batch(self):
        selected_keys = self.get_indices()
        B = self.batch_size
        pd, ph, pw = self.patch_size
        srcs_d, srcs_s = (C.c_void_p * B)(), (C.c_void_p * B)()
        shapes, lbs = (C.c_int * (3 * B))(), (C.c_int * (3 * B))()
        have_seg = True
        for j, k in enumerate(selected_keys):
            data, seg, _, props = self._data.load_case(k)
            force_fg = self.get_do_oversample(j) if seg is not None else False
            shape = tuple(data.shape[1:])
            bbox_lbs, _ = self.get_bbox(shape, force_fg, props.get('class_locations'))
            srcs_d[j] = data.data_ptr()
            have_seg &= seg is not None
            srcs_s[j] = seg.data_ptr() if seg is not None else None
            for a in range(3):
                shapes[3 * j + a], lbs[3 * j + a] = int(shape[a]), int(bbox_lbs[a])
        flips = (C.c_int * B)(*self._draw_flips())
        dev = self._data.device
        data_all = torch.empty((B, self.num_channels, pd, ph, pw), dtype=torch.float32, device=dev)
        call("nnz_crop_pad_f32", srcs_d, shapes, lbs, flips, ptr(data_all), B, self.num_channels, pd, ph, pw, 0.0,
             stream_ptr())
        seg_all = None
        if have_seg:
            seg_all = torch.empty((B, 1, pd, ph, pw), dtype=torch.int16, device=dev)
            call("nnz_crop_pad_i16", srcs_s, shapes, lbs, flips, ptr(seg_all), B, 1, pd, ph, pw, -1, stream_ptr())
        if self.patch_size_was_2d:
            data_all = data_all[:, :, 0]
            seg_all = seg_all[:, :, 0] if seg_all is not None else None
        if self.augmenter is not None:
            data_all, seg_all = self.augmenter(data_all.contiguous(), seg_all.contiguous() if seg_all is not None else None)
        if seg_all is not None and self.deep_supervision_scales is not None:
            seg_all = downsample_seg_for_ds(seg_all, self.deep_supervision_scales)
        batch = {'data': data_all, 'target': seg_all, 'keys': selected_keys}
        if self.transforms is not None:
            batch = self.transforms(batch)
        return batch

    def __iter__(self):
        return self

    def __next__(self):
        return self.generate_train_batch()

    def __len__(self):
        return int(np.ceil(len(self.indices) / self.batch_size))


def downsample_seg_for_ds(seg: torch.Tensor, ds_scales) -> List[torch.Tensor]:
    """DownsampleSegForDSTransform (nnUNetTrainer.py:971; batchgeneratorsv2, absent: restated - one target per scale,
    scale 1 = the tensor itself, otherwise interpolate(..., size=round(shape * scale), mode='nearest-exact'))"""
    if not seg.is_cuda or seg.dtype != torch.int16:
        raise RuntimeError("downsample_seg_for_ds: int16 CUDA tensor expected (no CPU path)")
    sp = tuple(seg.shape[2:])
    nd = len(sp)
    out = []
    for s in ds_scales:
        if all(i == 1 for i in s):
            out.append(seg)
            continue
        new = tuple(int(round(i * j)) for i, j in zip(sp, s))
        dst = torch.empty((*seg.shape[:2], *new), dtype=torch.int16, device=seg.device)
        i3 = (1,) * (3 - nd) + sp
        o3 = (1,) * (3 - nd) + new
        call("nnz_downsample_nearest_i16", ptr(seg.contiguous()), ptr(dst), seg.shape[0] * seg.shape[1], *i3, *o3, stream_ptr())
        out.append(dst)
    return out
