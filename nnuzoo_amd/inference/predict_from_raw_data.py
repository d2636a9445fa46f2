"""Sliding-window predictor for MI355X: the second caller of the network hot path.

Mirrors the prediction core of /root/reference/nnunetv2/inference/predict_from_raw_data.py:
  nnUNetPredictor.__init__ :37-66, manual_initialization :139-173, predict_logits_from_preprocessed_data :469-517,
  _internal_get_sliding_window_slicers :519-547, _internal_maybe_mirror_and_predict :549-564,
  _internal_predict_sliding_window_return_logits :566-643, predict_sliding_window_return_logits :645-692.
File IO, preprocessing, resampling/export and the multi-process pipeline around it are out of scope.

MI355X-first differences (results are bit-identical to the reference loop given the same network outputs):
  * all mirror variants of a tile go through the network as ONE batch (1 + 7 forwards of batch 1 -> one of batch 8),
    and several tiles per forward when mirroring is off (`tiles_per_forward`);
  * mirror merge (`+= flip(...)`, `/= n`), `*= gaussian`, `logits[sl] +=` and `n_predictions[sl] +=` are one HIP
    kernel per tile (csrc/sliding_window.hip) that rounds to fp16 exactly where the reference's half tensors do; the
    final `/= n_predictions` + inf check is a second kernel;
  * results always live on the device (288 GB of HBM: a 1000^3 x 4-class fp16 logit volume is 8 GB); there is no CPU
    accumulation path - a CPU device raises.
"""
from __future__ import annotations

import itertools
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import nn

from .._lib import call, ptr, stream_ptr
from .sliding_window_prediction import compute_gaussian, compute_steps_for_sliding_window, pad_to_tile

import ctypes as C


def _int3(v: Sequence[int]):
    arr = (C.c_int * 3)()
    for i, x in enumerate(v):
        arr[i] = int(x)
    return arr


class nnUNetPredictor(object):
    def __init__(self,
                 tile_step_size: float = 0.5,
                 use_gaussian: bool = True,
                 use_mirroring: bool = True,
                 perform_everything_on_device: bool = True,
                 device: torch.device = torch.device('cuda'),
                 verbose: bool = False,
                 verbose_preprocessing: bool = False,
                 allow_tqdm: bool = True,
                 tiles_per_forward: int = 4):
        self.verbose, self.verbose_preprocessing, self.allow_tqdm = verbose, verbose_preprocessing, allow_tqdm
        self.plans_manager = self.configuration_manager = self.list_of_parameters = self.network = None
        self.dataset_json = self.trainer_name = self.allowed_mirroring_axes = self.label_manager = None
        self.target_type = None
        self.tile_step_size = tile_step_size
        self.use_gaussian = use_gaussian
        self.use_mirroring = use_mirroring
        device = torch.device(device)
        if device.type != 'cuda':
            raise RuntimeError("nnuzoo_amd.nnUNetPredictor runs on MI355X through libnnuzoo_hip.so only; there is "
                               "deliberately no CPU path (see oracle/sliding_window.py for the test-only restatement)")
        self.device = device
        self.perform_everything_on_device = True
        self.tiles_per_forward = max(1, int(tiles_per_forward))

    # ---- reference API: used by nnUNetTrainer.perform_actual_validation and by callers that build the net themselves
    def manual_initialization(self, network: nn.Module, plans_manager, configuration_manager,
                              parameters: Optional[List[dict]], dataset_json: dict, trainer_name: str,
                              inference_allowed_mirroring_axes: Optional[Tuple[int, ...]], label_manager=None):
        self.plans_manager = plans_manager
        self.configuration_manager = configuration_manager
        self.list_of_parameters = parameters
        self.network = network
        self.dataset_json = dataset_json
        self.target_type = (dataset_json or {}).get('target_type', 'segmentation')
        self.trainer_name = trainer_name
        self.allowed_mirroring_axes = inference_allowed_mirroring_axes
        if label_manager is None and plans_manager is not None and hasattr(plans_manager, 'get_label_manager'):
            label_manager = plans_manager.get_label_manager(dataset_json)
        self.label_manager = label_manager

    @property
    def _patch_size(self) -> Tuple[int, ...]:
        return tuple(int(i) for i in self.configuration_manager.patch_size)

    @property
    def _num_heads(self) -> int:
        return int(self.label_manager.num_segmentation_heads)

    # ---- predict_from_raw_data.py:469-517 ------------------------------------------------------------------------
    @torch.inference_mode()
    def predict_logits_from_preprocessed_data(self, data: torch.Tensor) -> torch.Tensor:
        prediction = None
        params = self.list_of_parameters if self.list_of_parameters else [None]
        for p in params:
            if p is not None:
                self.network.load_state_dict(p)
            cur = self.predict_sliding_window_return_logits(data).to('cpu')
            prediction = cur if prediction is None else prediction + cur
        if len(params) > 1:
            prediction /= len(params)
        return prediction

    # ---- :519-547 ------------------------------------------------------------------------------------------------
    def _internal_get_sliding_window_slicers(self, image_size: Tuple[int, ...]):
        patch = self._patch_size
        slicers = []
        if len(patch) < len(image_size):
            assert len(patch) == len(image_size) - 1, \
                'if tile_size has less entries than image_size, len(tile_size) must be one shorter than ' \
                'len(image_size) (only dimension discrepancy of 1 allowed).'
            steps = compute_steps_for_sliding_window(image_size[1:], patch, self.tile_step_size)
            for d in range(image_size[0]):
                for sx in steps[0]:
                    for sy in steps[1]:
                        slicers.append(tuple([slice(None), d, *[slice(si, si + ti) for si, ti in zip((sx, sy), patch)]]))
        else:
            steps = compute_steps_for_sliding_window(image_size, patch, self.tile_step_size)
            for origin in itertools.product(*steps):
                slicers.append(tuple([slice(None), *[slice(si, si + ti) for si, ti in zip(origin, patch)]]))
        return slicers

    # ---- :549-564, restated as data: which flips, in the reference's accumulation order --------------------------
    def _mirror_axes_combinations(self, spatial_ndim: int) -> List[Tuple[int, ...]]:
        """[()] + the reference's `axes_combinations` (as spatial axis indices), in its order"""
        mirror_axes = self.allowed_mirroring_axes if self.use_mirroring else None
        combos: List[Tuple[int, ...]] = [()]
        if mirror_axes is not None:
            assert max(mirror_axes) <= spatial_ndim - 1, 'mirror_axes does not match the dimension of the input!'
            for i in range(len(mirror_axes)):
                combos += list(itertools.combinations(list(mirror_axes), i + 1))
        return combos

    def _forward_logits(self, x: torch.Tensor) -> torch.Tensor:
        if hasattr(self.network, "grad_arena") or not x.is_cuda:
            out = self.network(x)  # native HIP schedule: autocast numerics are built in
        else:
            # any other network class named in plans.json runs under torch.autocast exactly as in the reference
            # (predict_from_raw_data.py:586 wraps the whole loop in `torch.autocast(device.type, enabled=True)`)
            with torch.autocast("cuda", enabled=True):
                out = self.network(x)
        if isinstance(out, (list, tuple)):
            out = out[0]
        if out.dtype != torch.float16:
            # the reference runs the network under torch.autocast: conv outputs (the logits) are half tensors
            out = out.to(torch.float16)
        return out.contiguous()

    def _internal_maybe_mirror_and_predict(self, x: torch.Tensor) -> torch.Tensor:
        """Reference-shaped helper (one tile batch in, mirror-averaged fp16 logits out).  The sliding-window loop does
        not call it: it feeds the un-merged mirror batch to the accumulation kernel instead."""
        nsp = x.dim() - 2
        combos = self._mirror_axes_combinations(nsp)
        logits = None
        for b in range(x.shape[0]):
            batch = torch.cat([torch.flip(x[b:b + 1], [a + 2 for a in c]) if c else x[b:b + 1] for c in combos])
            out = self._forward_logits(batch)
            K, tile = out.shape[1], tuple(out.shape[2:])
            if logits is None:
                logits = torch.zeros((x.shape[0], K, *tile), dtype=torch.float16, device=x.device)
            npred = torch.zeros(tile, dtype=torch.float16, device=x.device)
            self._accumulate(out, combos, None, logits[b], npred, (0,) * nsp)
        return logits

    # ---- the fused accumulation ------------------------------------------------------------------------------------
    @staticmethod
    def _flip_bits(combos: Sequence[Tuple[int, ...]], nsp: int):
        arr = (C.c_int * 8)()
        for m, c in enumerate(combos):
            bits = 0
            for a in c:
                bits |= 1 << (a + (3 - nsp))  # 2-D tiles are depth-1 volumes: spatial axis a -> volume axis a + 1
            arr[m] = bits
        return arr

    def _accumulate(self, preds: torch.Tensor, combos, gaussian: Optional[torch.Tensor], logits: torch.Tensor,
                    npred: torch.Tensor, origin: Sequence[int]):
        """preds [M, K, *tile] fp16 (M = len(combos)); logits [K, *image], npred [*image] fp16, contiguous.
        origin has one entry per image axis; a 2-D tile inside a 3-D image is a depth-1 tile at origin (d, y, x)."""
        nsp = preds.dim() - 2
        if len(combos) > 8:
            raise ValueError("at most 3 mirror axes")
        image = tuple(logits.shape[1:])
        image = (1,) * (3 - len(image)) + image
        tile = (1,) * (3 - nsp) + tuple(preds.shape[2:])
        off = tuple(int(o) for o in origin)
        off = (0,) * (3 - len(off)) + off
        assert logits.is_contiguous() and npred.is_contiguous() and preds.is_contiguous()
        call("nnz_sliding_window_accumulate", ptr(preds), len(combos), self._flip_bits(combos, nsp), ptr(gaussian),
             ptr(logits), ptr(npred), int(preds.shape[1]), _int3(tile), _int3(image), _int3(off), stream_ptr())

    # ---- :566-643 ------------------------------------------------------------------------------------------------
    def _internal_predict_sliding_window_return_logits(self, data: torch.Tensor, slicers, do_on_device: bool = True):
        dev = self.device
        data = data.to(dev)
        if data.dtype not in (torch.float32, torch.float16):
            data = data.float()
        K = self._num_heads
        predicted_logits = torch.zeros((K, *data.shape[1:]), dtype=torch.half, device=dev)
        n_predictions = torch.zeros(data.shape[1:], dtype=torch.half, device=dev)
        gaussian = compute_gaussian(self._patch_size, sigma_scale=1. / 8, value_scaling_factor=10, device=dev) \
            if self.use_gaussian else None
        nsp = len(self._patch_size)
        combos = self._mirror_axes_combinations(nsp)
        per_fwd = self.tiles_per_forward if len(combos) == 1 else 1  # mirroring already fills the batch dimension
        for i0 in range(0, len(slicers), per_fwd):
            group = slicers[i0:i0 + per_fwd]
            tiles = []
            for sl in group:
                w = data[sl][None]
                tiles += [torch.flip(w, [a + 2 for a in c]) if c else w for c in combos]
            out = self._forward_logits(torch.cat(tiles).float())  # [len(group) * M, K, *tile]
            M = len(combos)
            for j, sl in enumerate(group):
                # origin per image axis; a 2-D network on a 3-D image addresses slice d: depth-1 tile at (d, y, x)
                origin = [s_.start if isinstance(s_, slice) else int(s_) for s_ in sl[1:]]
                self._accumulate(out[j * M:(j + 1) * M], combos, gaussian, predicted_logits, n_predictions, origin)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        V = int(np.prod(data.shape[1:]))
        call("nnz_sliding_window_finalize", ptr(predicted_logits), ptr(n_predictions), K, V, ptr(flag), stream_ptr())
        if int(flag.item()):
            raise RuntimeError('Encountered inf in predicted array. Aborting... If this problem persists, '
                               'reduce value_scaling_factor in compute_gaussian or increase the dtype of '
                               'predicted_logits to fp32')
        return predicted_logits

    # ---- :645-692 ------------------------------------------------------------------------------------------------
    @torch.inference_mode()
    def predict_sliding_window_return_logits(self, input_image: torch.Tensor) -> torch.Tensor:
        assert isinstance(input_image, torch.Tensor)
        assert input_image.ndim == 4, 'input_image must be a 4D np.ndarray or torch.Tensor (c, x, y, z)'
        self.network = self.network.to(self.device)
        self.network.eval()
        if hasattr(self.network, 'decoder') and hasattr(self.network.decoder, 'deep_supervision'):
            self.network.decoder.deep_supervision = False
        data, slicer_revert_padding = pad_to_tile(input_image, self._patch_size)
        slicers = self._internal_get_sliding_window_slicers(data.shape[1:])
        predicted_logits = self._internal_predict_sliding_window_return_logits(data, slicers, True)
        return predicted_logits[tuple([slice(None), *slicer_revert_padding[1:]])]
