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
DownsampleSegForDSTransform.  The intensity / spatial augmentations (batchgeneratorsv2, absent from the reference tree
and from this image: PARITY UNPINNED, not built) can be appended as `transforms`, a callable on the device batch.

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


class nnUNetDataLoader:
    def __init__(self, data: DeviceCaseStore, batch_size: int, initial_patch_size, final_patch_size, label_manager,
                 oversample_foreground_percent: float = 0.0, sampling_probabilities=None, pad_sides=None,
                 probabilistic_oversampling: bool = False, transforms=None, target_type: str = "segmentation",
                 deep_supervision_scales: Optional[Sequence[Sequence[float]]] = None,
                 mirror_axes: Optional[Tuple[int, ...]] = None):
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
        """data_loader.py:102-178, same statements in the same order (the numpy RNG stream is part of the contract)"""
        need_to_pad = self.need_to_pad.copy()
        dim = len(data_shape)
        for d in range(dim):
            if need_to_pad[d] + data_shape[d] < self.patch_size[d]:
                need_to_pad[d] = self.patch_size[d] - data_shape[d]
        lbs = [- need_to_pad[i] // 2 for i in range(dim)]
        ubs = [data_shape[i] + need_to_pad[i] // 2 + need_to_pad[i] % 2 - self.patch_size[i] for i in range(dim)]
        if not force_fg and not self.has_ignore:
            bbox_lbs = [np.random.randint(lbs[i], ubs[i] + 1) for i in range(dim)]
        else:
            if not force_fg and self.has_ignore:
                selected_class = self.annotated_classes_key
                if len(class_locations[selected_class]) == 0:
                    warnings.warn('Warning! No annotated pixels in image!')
                    selected_class = None
            elif force_fg:
                assert class_locations is not None, 'if force_fg is set class_locations cannot be None'
                if overwrite_class is not None:
                    assert overwrite_class in class_locations.keys(), \
                        'desired class ("overwrite_class") does not have class_locations (missing key)'
                eligible = [i for i in class_locations.keys() if len(class_locations[i]) > 0]
                tmp = [i == self.annotated_classes_key if isinstance(i, tuple) else False for i in eligible]
                if any(tmp):
                    if len(eligible) > 1:
                        eligible.pop(np.where(tmp)[0][0])
                if len(eligible) == 0:
                    selected_class = None
                else:
                    selected_class = eligible[np.random.choice(len(eligible))] if \
                        (overwrite_class is None or (overwrite_class not in eligible)) else overwrite_class
            else:
                raise RuntimeError('lol what!?')
            if selected_class is not None:
                voxels = class_locations[selected_class]
                selected_voxel = voxels[np.random.choice(len(voxels))]
                bbox_lbs = [max(lbs[i], selected_voxel[i + 1] - self.patch_size[i] // 2) for i in range(dim)]
            else:
                bbox_lbs = [np.random.randint(lbs[i], ubs[i] + 1) for i in range(dim)]
        bbox_ubs = [bbox_lbs[i] + self.patch_size[i] for i in range(dim)]
        return bbox_lbs, bbox_ubs

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
