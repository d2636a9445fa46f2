"""Synthetic patches and planner output for the benchmark configurations (BASELINE.json `configs`).

* `nnunet_plans(dim, patch)` reproduces what the reference's planner emits for an isotropic patch:
  topology rule of /root/reference/nnunetv2/experiment_planning/experiment_planners/network_topology.py:30-105
  (halve every axis while the feature map edge stays >= 4 (UNet_featuremap_min_edge_length), kernels 3, first
  stride 1) and the arch kwargs of default_experiment_planner.py:285-305 (features min(32 * 2^i, 320 in 3-D /
  512 in 2-D), 2 convs per stage, InstanceNorm(eps 1e-5, affine), LeakyReLU(inplace), conv_bias).
  128^3 -> 6 stages [32, 64, 128, 256, 320, 320]; 512^2 -> 8 stages [32, ..., 512, 512, 512, 512].
* `synthetic_batch` is the seeded generator of SURVEY.md §8d: z-scored noise image, one random axis-aligned
  ellipsoid per sample as foreground (+1.0 intensity inside), int16 deep-supervision targets by nearest-neighbour
  subsampling - the batch dict layout of nnUNetDataLoader.generate_train_batch (dataloading/data_loader.py:259).
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch


def nnunet_plans(dim: int, patch: Sequence[int], batch_size: int = 2, num_classes: int = 2) -> tuple:
    min_edge = 4
    max_feat = 320 if dim == 3 else 512
    edges = list(patch)
    strides = [[1] * dim]
    while all(e // 2 >= min_edge and e % 2 == 0 for e in edges):
        edges = [e // 2 for e in edges]
        strides.append([2] * dim)
    n_stages = len(strides)
    conv = "torch.nn.modules.conv.Conv%dd" % dim
    norm = "torch.nn.modules.instancenorm.InstanceNorm%dd" % dim
    arch = {
        'network_class_name': 'dynamic_network_architectures.architectures.unet.PlainConvUNet',
        'arch_kwargs': {
            'n_stages': n_stages,
            'features_per_stage': [min(32 * 2 ** i, max_feat) for i in range(n_stages)],
            'conv_op': conv,
            'kernel_sizes': [[3] * dim for _ in range(n_stages)],
            'strides': strides,
            'n_conv_per_stage': [2] * n_stages,
            'n_conv_per_stage_decoder': [2] * (n_stages - 1),
            'conv_bias': True,
            'norm_op': norm,
            'norm_op_kwargs': {'eps': 1e-5, 'affine': True},
            'dropout_op': None,
            'dropout_op_kwargs': None,
            'nonlin': 'torch.nn.LeakyReLU',
            'nonlin_kwargs': {'inplace': True},
        },
        '_kw_requires_import': ('conv_op', 'norm_op', 'dropout_op', 'nonlin'),
    }
    cfg_name = '3d_fullres' if dim == 3 else '2d'
    plans = {'configurations': {cfg_name: {'patch_size': list(patch), 'batch_size': batch_size,
                                           'batch_dice': dim == 2, 'architecture': arch}}}
    dataset_json = {'channel_names': {'0': 'synthetic'},
                    'labels': {'background': 0, **{f'fg{i}': i for i in range(1, num_classes)}}}
    return plans, cfg_name, dataset_json


def synthetic_batch(batch: int, patch: Sequence[int], ds_scales: List[List[float]], seed: int = 1234) -> dict:
    g = torch.Generator().manual_seed(seed)
    dim = len(patch)
    data = torch.randn(batch, 1, *patch, generator=g)
    grids = torch.meshgrid(*[torch.arange(s, dtype=torch.float32) for s in patch], indexing='ij')
    seg = torch.zeros(batch, 1, *patch, dtype=torch.int16)
    for b in range(batch):
        centre = (0.3 + 0.4 * torch.rand(dim, generator=g)) * torch.tensor(patch, dtype=torch.float32)
        radii = (0.1 + 0.2 * torch.rand(dim, generator=g)) * torch.tensor(patch, dtype=torch.float32)
        d2 = sum(((grids[a] - centre[a]) / radii[a]) ** 2 for a in range(dim))
        fg = d2 <= 1.0
        seg[b, 0][fg] = 1
        data[b, 0][fg] += 1.0
    targets = []
    for sc in ds_scales:
        steps = [int(round(1 / s)) for s in sc]
        sl = (slice(None), slice(None)) + tuple(slice(None, None, st) for st in steps)
        targets.append(seg[sl].contiguous())
    return {'data': data, 'target': targets, 'keys': [f'synthetic_{seed}_{b}' for b in range(batch)]}


def conv_flops_forward(arch_kwargs: dict, patch: Sequence[int], in_ch: int = 1, num_classes: int = 2) -> dict:
    """2*MAC conv FLOPs of one forward sample, per layer (SURVEY.md §8d formulas)."""
    feats = arch_kwargs['features_per_stage']
    strides = arch_kwargs['strides']
    dim = len(patch)
    k = 3 ** dim
    out = {}
    edges = list(patch)
    cin = in_ch
    lvl_vox = []
    for s, f in enumerate(feats):
        edges = [e // st for e, st in zip(edges, strides[s])]
        vox = int(np.prod(edges))
        lvl_vox.append(vox)
        for i in range(arch_kwargs['n_conv_per_stage'][s]):
            out[f'enc{s}.{i}'] = 2.0 * vox * cin * f * k
            cin = f
    S = len(feats)
    for lvl in range(S - 2, -1, -1):
        below, skip = feats[lvl + 1], feats[lvl]
        out[f'up{lvl}'] = 2.0 * lvl_vox[lvl + 1] * below * skip * (2 ** dim)
        c = 2 * skip
        for i in range(arch_kwargs['n_conv_per_stage_decoder'][lvl]):
            out[f'dec{lvl}.{i}'] = 2.0 * lvl_vox[lvl] * c * skip * k
            c = skip
        out[f'seg{lvl}'] = 2.0 * lvl_vox[lvl] * skip * num_classes
    return out
