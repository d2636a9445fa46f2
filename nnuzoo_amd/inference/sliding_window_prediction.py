"""Tile geometry and importance map of the sliding-window predictor.

Same public functions as /root/reference/nnunetv2/inference/sliding_window_prediction.py (`compute_gaussian` :10-29,
`compute_steps_for_sliding_window` :32-58); host-side set-up code that runs once per image shape.
"""
from __future__ import annotations

from functools import lru_cache
from typing import List, Sequence, Tuple, Union

import numpy as np
import torch
from scipy.ndimage import gaussian_filter


@lru_cache(maxsize=4)
def compute_gaussian(tile_size: Union[Tuple[int, ...], List[int]], sigma_scale: float = 1. / 8,
                     value_scaling_factor: float = 1, dtype=torch.float16, device=torch.device('cuda', 0)) \
        -> torch.Tensor:
    """Importance map: a unit impulse at the tile centre blurred with sigma = tile * sigma_scale per axis, scaled so
    that its maximum is `value_scaling_factor`; zeros (fp16 underflow in the corners) are lifted to the smallest
    non-zero entry so that the final division never sees 0."""
    impulse = np.zeros(tile_size)
    impulse[tuple(i // 2 for i in tile_size)] = 1
    g = gaussian_filter(impulse, [i * sigma_scale for i in tile_size], 0, mode='constant', cval=0)
    g = torch.from_numpy(g)
    g = (g / torch.max(g) * value_scaling_factor).type(dtype).to(device)
    g[g == 0] = torch.min(g[g != 0])
    return g


def compute_steps_for_sliding_window(image_size: Sequence[int], tile_size: Sequence[int], tile_step_size: float) \
        -> List[List[int]]:
    """Tile origins per axis: at most tile*step apart, evenly spread so that first / last tile touch the borders
    (image 110, tile 64, step 0.5 -> 0, 23, 46)."""
    assert all(i >= j for i, j in zip(image_size, tile_size)), "image size must be as large or larger than patch_size"
    assert 0 < tile_step_size <= 1, 'step_size must be larger than 0 and smaller or equal to 1'
    steps = []
    for img, tile in zip(image_size, tile_size):
        n = int(np.ceil((img - tile) / (tile * tile_step_size))) + 1
        last = img - tile
        stride = last / (n - 1) if n > 1 else 99999999999  # one tile at 0: the stride is irrelevant
        steps.append([int(np.round(stride * i)) for i in range(n)])
    return steps


def pad_to_tile(image: torch.Tensor, tile_size: Sequence[int]):
    """Zero-pad the trailing len(tile_size) axes of `image` up to the tile size, centred (the odd voxel goes to the
    upper side), and return (padded, slicer that crops a same-rank array back).  Behaviour of acvl_utils.pad_nd_image
    with mode 'constant' / value 0 / return_slicer=True as called at predict_from_raw_data.py:668-670."""
    nd = len(tile_size)
    lead = image.dim() - nd
    old = list(image.shape[lead:])
    new = [max(o, int(t)) for o, t in zip(old, tile_size)]
    below = [(n - o) // 2 for n, o in zip(new, old)]
    above = [(n - o) - b for n, o, b in zip(new, old, below)]
    slicer = tuple([slice(None)] * lead + [slice(b, b + o) for b, o in zip(below, old)])
    if not any(below) and not any(above):
        return image, slicer
    pads = []
    for b, a in zip(reversed(below), reversed(above)):  # F.pad takes the last axis first
        pads += [b, a]
    return torch.nn.functional.pad(image, pads, mode='constant', value=0), slicer
