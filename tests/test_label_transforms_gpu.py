"""Label-side transforms of the training chain on the resident batch (csrc/augment.hip: region maps, one-hot move of the cascade,
normalisation masks) against what the REFERENCE's own classes return (tests/golden/label_transforms.npz, written by
tools/make_golden_label_transforms.py from /root/reference/nnunetv2/training/data_augmentation/custom_transforms/*.py):
bit-exact (integer / 0-1 work)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "label_transforms.npz"))
REGIONS = ((1, 2, 3), (2, 3), 3, (4,), -1)


@pytest.mark.parametrize("tag", ["2d", "3d"])
def test_regions_onehot_mask_match_the_reference_classes(hip_lib, tag):
    from nnuzoo_amd.dataloading.device_augment import mask_outside, move_seg_as_onehot_to_data, seg_to_regions
    seg = torch.from_numpy(Z[f"seg_{tag}"]).cuda()
    data = torch.from_numpy(Z[f"data_{tag}"]).cuda()
    got = seg_to_regions(seg, REGIONS, 0)
    assert got.dtype == torch.int16 and np.array_equal(got.cpu().numpy(), Z[f"regions_{tag}"])
    d, s = move_seg_as_onehot_to_data(data.clone(), seg.clone(), 1, (1, 2, 4), remove_from_origin=True)
    assert np.array_equal(d.cpu().numpy(), Z[f"onehot_data_{tag}"]) and np.array_equal(s.cpu().numpy(), Z[f"onehot_seg_{tag}"])
    m = mask_outside(data.clone(), seg, [1], 0, 0.0)
    assert np.array_equal(m.cpu().numpy(), Z[f"masked_{tag}"])


def test_augmenter_applies_them_in_the_chain_order(hip_lib):
    """mask -> remove label -1 -> cascade one-hot -> regions, with every random transform switched off"""
    from nnuzoo_amd.dataloading.device_augment import DeviceAugmenter
    seg = torch.from_numpy(Z["seg_2d"]).cuda()
    data = torch.from_numpy(Z["data_2d"]).cuda()
    aug = DeviceAugmenter(seg.shape[2:], (-0.1, 0.1), seed=1, use_mask_for_norm=[False, True], cascade_labels=(1, 2, 4),
                          regions=((1, 2, 3), (2, 3), 3, (4,)), ignore_label=None)
    aug.p_rotation = aug.p_scaling = aug.p_noise = aug.p_blur = aug.p_brightness = aug.p_contrast = aug.p_lowres = 0.0
    aug.p_gamma = aug.p_gamma_inverted = 0.0
    d, s = aug(data.clone(), seg.clone())
    want_d = Z["data_2d"].copy()
    want_d[:, 1][Z["seg_2d"][:, 0] < 0] = 0
    sg = Z["seg_2d"].copy()
    sg[sg == -1] = 0
    onehot = np.stack([(sg[:, 1] == l).astype(np.float32) for l in (1, 2, 4)], 1)
    assert np.array_equal(d.cpu().numpy(), np.concatenate([want_d, onehot], 1))
    reg = np.stack([np.isin(sg[:, 0], r if isinstance(r, tuple) else (r,)).astype(np.int16) for r in ((1, 2, 3), (2, 3), 3, (4,))], 1)
    assert np.array_equal(s.cpu().numpy(), reg)
