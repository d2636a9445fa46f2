"""GPU-side input pipeline (nnuzoo_amd/dataloading/device_loader.py, csrc/input_pipeline.hip).

Pinned: the bounding-box rule - `get_bbox` against tests/golden/dataloader_bbox.json, outputs of the reference's own
method (data_loader.py:102-178) extracted by ast and executed unchanged (tools/make_golden.py gen_dataloader_bbox), same
numpy seed -> same boxes.  Bit-exact by construction (integer / copy work): crop + pad (+ mirror) against a numpy
restatement of crop_and_pad_nd's contract (acvl_utils, absent: unpinned), deep-supervision targets against torch's own
interpolate(mode='nearest-exact')."""
import json
import os
import types

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden", "dataloader_bbox.json")


def _class_locations(sc):
    """restated from tools/make_golden.py (the fixture stores the parameters, not the arrays)"""
    if sc["classes"] == "none":
        return None
    rs = np.random.RandomState(100 + sc["seed"])
    sh = sc["data_shape"]

    def locs(n):
        return np.stack([np.zeros(n, dtype=np.int64)] + [rs.randint(0, s, n) for s in sh], 1)

    if sc["classes"] == "two":
        return {1: locs(40), 2: locs(7), 3: np.zeros((0, 4), dtype=np.int64)}
    if sc["classes"] == "ignore":
        return {1: locs(30), 2: np.zeros((0, 4), dtype=np.int64), tuple(sc["annotated_classes_key"]): locs(60)}
    return {1: np.zeros((0, 4), dtype=np.int64), 2: np.zeros((0, 4), dtype=np.int64)}


def test_get_bbox_matches_the_reference_function():
    from nnuzoo_amd.dataloading.device_loader import nnUNetDataLoader
    for sc in json.load(open(G)):
        self_ = types.SimpleNamespace(need_to_pad=np.array(sc["need_to_pad"]), patch_size=tuple(sc["patch_size"]),
                                      has_ignore=sc["has_ignore"], annotated_classes_key=tuple(sc["annotated_classes_key"]))
        np.random.seed(sc["seed"])
        cl = _class_locations(sc)
        for i, want in enumerate(sc["boxes"]):
            lbs, ubs = nnUNetDataLoader.get_bbox(self_, np.array(sc["data_shape"]), sc["force_fg"][i % len(sc["force_fg"])], cl)
            assert [[int(v) for v in lbs], [int(v) for v in ubs]] == want, (sc["seed"], i)


def test_host_rules():
    from nnuzoo_amd.dataloading.device_loader import nnUNetDataLoader
    s = types.SimpleNamespace(batch_size=4, oversample_foreground_percent=0.33)
    # data_loader.py:77-81: the LAST round(B * p) samples of a batch are forced foreground
    assert [nnUNetDataLoader._oversample_last_XX_percent(s, j) for j in range(4)] == [False, False, False, True]
    s.batch_size, s.oversample_foreground_percent = 2, 0.33
    assert [nnUNetDataLoader._oversample_last_XX_percent(s, j) for j in range(2)] == [False, True]
    with pytest.raises(RuntimeError):
        from nnuzoo_amd.dataloading.device_loader import DeviceCaseStore
        DeviceCaseStore(types.SimpleNamespace(identifiers=[]), device="cpu")


class _Cases:
    def __init__(self, shapes, channels=2, seed=0):
        rs = np.random.RandomState(seed)
        self.identifiers = [f"case_{i}" for i in range(len(shapes))]
        self.cases = {}
        for k, sh in zip(self.identifiers, shapes):
            data = rs.randn(channels, *sh).astype(np.float32)
            seg = rs.randint(0, 3, (1, *sh)).astype(np.int16)
            locs = {c: np.argwhere(seg == c) for c in (1, 2)}
            self.cases[k] = (data, seg, None, {"class_locations": locs})

    def load_case(self, k):
        return self.cases[k]


def _crop_pad_np(vol, lbs, patch, pad, flip):
    """crop_and_pad_nd's contract: the box [lb, lb + patch) of every channel, `pad` outside the case; then the mirror"""
    out = np.full((vol.shape[0], *patch), pad, dtype=vol.dtype)
    src, dst = [], []
    for a in range(3):
        lo, hi = max(lbs[a], 0), min(lbs[a] + patch[a], vol.shape[1 + a])
        src.append(slice(lo, hi))
        dst.append(slice(lo - lbs[a], hi - lbs[a]))
    if all(s.stop > s.start for s in src):
        out[(slice(None), *dst)] = vol[(slice(None), *src)]
    for a in range(3):
        if flip >> a & 1:
            out = np.flip(out, 1 + a)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("patch,shapes", [((32, 40, 24), [(40, 56, 48), (20, 30, 64), (33, 41, 25)]),
                                          ((1, 48, 64), [(1, 50, 40), (1, 96, 80)])])
def test_crop_pad_mirror_kernels_bit_exact(hip_lib, patch, shapes):
    import ctypes as C
    from nnuzoo_amd._lib import call, ptr, stream_ptr
    ds = _Cases(shapes)
    rs = np.random.RandomState(3)
    B = 19                                          # more than one launch group of 16
    keys = [ds.identifiers[i] for i in rs.randint(0, len(shapes), B)]
    dev = {k: (torch.tensor(v[0]).cuda(), torch.tensor(v[1]).cuda()) for k, v in ds.cases.items()}
    lbs = [[int(rs.randint(-patch[a], ds.cases[k][0].shape[1 + a])) for a in range(3)] for k in keys]   # incl. fully outside
    flips = [int(rs.randint(0, 8)) for _ in keys]
    sd, ss = (C.c_void_p * B)(), (C.c_void_p * B)()
    sh, lb = (C.c_int * (3 * B))(), (C.c_int * (3 * B))()
    for j, k in enumerate(keys):
        sd[j], ss[j] = dev[k][0].data_ptr(), dev[k][1].data_ptr()
        for a in range(3):
            sh[3 * j + a], lb[3 * j + a] = ds.cases[k][0].shape[1 + a], lbs[j][a]
    fl = (C.c_int * B)(*flips)
    out_d = torch.empty((B, 2, *patch), dtype=torch.float32, device="cuda")
    out_s = torch.empty((B, 1, *patch), dtype=torch.int16, device="cuda")
    call("nnz_crop_pad_f32", sd, sh, lb, fl, ptr(out_d), B, 2, *patch, 0.0, stream_ptr())
    call("nnz_crop_pad_i16", ss, sh, lb, fl, ptr(out_s), B, 1, *patch, -1, stream_ptr())
    for j, k in enumerate(keys):
        assert np.array_equal(out_d[j].cpu().numpy(), _crop_pad_np(ds.cases[k][0], lbs[j], patch, 0, flips[j])), j
        assert np.array_equal(out_s[j].cpu().numpy(), _crop_pad_np(ds.cases[k][1], lbs[j], patch, -1, flips[j])), j


@pytest.mark.gpu
@pytest.mark.parametrize("shape,scales", [((32, 40, 24), [[1, 1, 1], [0.5, 0.5, 0.5], [0.25, 0.5, 0.25], [0.125, 0.25, 0.125]]),
                                          ((96, 80), [[1, 1], [0.5, 0.5], [0.25, 0.25], [0.03125, 0.0625]]),
                                          ((37, 53), [[0.5, 0.5], [0.3, 0.7]])])
def test_deep_supervision_targets_equal_torch_nearest_exact(hip_lib, shape, scales):
    from nnuzoo_amd.dataloading.device_loader import downsample_seg_for_ds
    seg = torch.randint(-1, 5, (3, 1, *shape), dtype=torch.int16, device="cuda")
    outs = downsample_seg_for_ds(seg, scales)
    for s, o in zip(scales, outs):
        if all(i == 1 for i in s):
            assert o is seg
            continue
        new = [int(round(i * j)) for i, j in zip(shape, s)]
        ref = torch.nn.functional.interpolate(seg.float().cpu(), new, mode="nearest-exact").to(torch.int16)
        assert torch.equal(o.cpu(), ref), s


@pytest.mark.gpu
def test_loader_batches_and_feeds_a_train_step(hip_lib):
    """whole generate_train_batch against the same host draws replayed on numpy; then the batch drives a trainer step"""
    from nnuzoo_amd.dataloading.device_loader import DeviceCaseStore, nnUNetDataLoader
    from nnuzoo_amd.synthetic import nnunet_plans
    from nnuzoo_amd.training.nnUNetTrainer import nnUNetTrainer
    ds = _Cases([(40, 56, 48), (36, 36, 70), (50, 34, 34)], channels=1, seed=5)
    plans, cfg, dj = nnunet_plans(3, (32, 32, 32), batch_size=2)
    tr = nnUNetTrainer(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    scales = tr._get_deep_supervision_scales()
    lm = types.SimpleNamespace(all_labels=[0, 1, 2], has_ignore_label=False)
    store = DeviceCaseStore(ds)
    assert store.nbytes() == sum(v[0].nbytes + v[1].nbytes for v in ds.cases.values())
    dl = nnUNetDataLoader(store, 2, (40, 40, 40), (32, 32, 32), lm, oversample_foreground_percent=0.33,
                          deep_supervision_scales=scales, mirror_axes=(0, 1, 2))
    np.random.seed(11)
    batch = dl.generate_train_batch()
    # replay the host draws
    np.random.seed(11)
    keys = dl.get_indices()
    boxes = []
    for j, k in enumerate(keys):
        data, seg, _, props = ds.load_case(k)
        boxes.append(dl.get_bbox(data.shape[1:], dl.get_do_oversample(j), props["class_locations"])[0])
    flips = dl._draw_flips()
    assert list(batch["keys"]) == list(keys)
    assert batch["data"].shape == (2, 1, 32, 32, 32) and batch["data"].dtype == torch.float32
    assert [tuple(t.shape[2:]) for t in batch["target"]] == [tuple(int(round(32 * s)) for s in sc) for sc in scales]
    for j, k in enumerate(keys):
        assert np.array_equal(batch["data"][j].cpu().numpy(), _crop_pad_np(ds.cases[k][0], boxes[j], (32, 32, 32), 0, flips[j]))
        assert np.array_equal(batch["target"][0][j].cpu().numpy(), _crop_pad_np(ds.cases[k][1], boxes[j], (32, 32, 32), -1, flips[j]))
    # the forced-foreground sample (last of the batch) has its chosen voxel's class inside the patch
    assert (batch["target"][0][1] > 0).any()
    # targets may carry -1 (padding): nnU-Net maps it with the ignore / mask machinery upstream; clamp for this smoke step
    batch["target"] = [t.clamp_min(0) for t in batch["target"]]
    losses = [float(tr.train_step(dl_batch)["loss"]) for dl_batch in (batch, batch)]
    assert all(np.isfinite(losses))


@pytest.mark.gpu
def test_loader_with_the_device_augmenter_feeds_a_train_step(hip_lib):
    """the `augmenter` hook (dataloading/device_augment.DeviceAugmenter, csrc/augment.hip): augmented full-resolution batch, the
    deep-supervision targets cut AFTER it (they are down-samplings of the augmented segmentation), no -1 left in the targets"""
    from nnuzoo_amd.dataloading.device_augment import DeviceAugmenter
    from nnuzoo_amd.dataloading.device_loader import DeviceCaseStore, nnUNetDataLoader, downsample_seg_for_ds
    from nnuzoo_amd.synthetic import nnunet_plans
    from nnuzoo_amd.training.nnUNetTrainer import nnUNetTrainer
    ds = _Cases([(40, 56, 48), (36, 36, 70), (50, 34, 34)], channels=1, seed=5)
    plans, cfg, dj = nnunet_plans(3, (32, 32, 32), batch_size=2)
    tr = nnUNetTrainer(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    scales = tr._get_deep_supervision_scales()
    lm = types.SimpleNamespace(all_labels=[0, 1, 2], has_ignore_label=False)
    aug = DeviceAugmenter((32, 32, 32), (-0.5236, 0.5236), seed=3)          # rotation_for_DA of the 3-D configuration: +-30 degrees
    aug.p_rotation = aug.p_scaling = aug.p_gamma = 1.0                        # make sure the resampling and a statistics pass run
    dl = nnUNetDataLoader(DeviceCaseStore(ds), 2, (40, 40, 40), (32, 32, 32), lm, oversample_foreground_percent=0.33,
                          deep_supervision_scales=scales, mirror_axes=(0, 1, 2), augmenter=aug)
    np.random.seed(11)
    batch = dl.generate_train_batch()
    assert batch["data"].shape == (2, 1, 32, 32, 32) and torch.isfinite(batch["data"]).all()
    assert all(int((t < 0).sum()) == 0 for t in batch["target"])
    again = downsample_seg_for_ds(batch["target"][0], scales)
    assert all(torch.equal(a, b) for a, b in zip(again, batch["target"]))
    assert all(m is not None for m in aug.last["matrices"])
    losses = [float(tr.train_step(batch)["loss"]) for _ in range(2)]
    assert all(np.isfinite(losses))


def test_augmenter_host_draws():
    """the host half of DeviceAugmenter (no GPU): drawn maps are rotation x isotropic scale inside the call site's ranges
    (nnUNetTrainer.py:845-852), 2-D / dummy-2-D patches only turn in-plane, BGContrast ranges split at 1, draws repeat per seed"""
    from nnuzoo_amd.dataloading.device_augment import DeviceAugmenter
    a = DeviceAugmenter((32, 32, 32), (-0.5236, 0.5236), seed=1)
    a.p_rotation = a.p_scaling = 1.0
    for _ in range(50):
        m = a._draw_matrix(3)
        g = m @ m.T                                    # (R S)(R S)^T = s^2 I
        s2 = g[0, 0]
        assert np.allclose(g, s2 * np.eye(3), atol=1e-9) and 0.7 ** 2 - 1e-9 <= s2 <= 1.4 ** 2 + 1e-9
        assert np.linalg.det(m) > 0
    b = DeviceAugmenter((64, 64), (-3.14159, 3.14159), seed=2)
    b.p_rotation = b.p_scaling = 1.0
    for dummy in (False, True):
        b.dummy_2d = dummy
        m = b._draw_matrix(3 if dummy else 2)
        assert m[0, 0] == 1.0 and np.all(m[0, 1:] == 0) and np.all(m[1:, 0] == 0)       # the z axis is left alone
        assert 0.7 - 1e-9 <= np.sqrt(np.linalg.det(m[1:, 1:])) <= 1.4 + 1e-9
    c = DeviceAugmenter((8, 8), (0, 0), seed=3)
    draws = np.array([c._bg_range((0.75, 1.25)) for _ in range(4000)])
    assert draws.min() >= 0.75 and draws.max() <= 1.25 and abs((draws < 1).mean() - 0.5) < 0.04
    d1, d2 = DeviceAugmenter((8, 8), (-1, 1), seed=9), DeviceAugmenter((8, 8), (-1, 1), seed=9)
    d1.p_rotation = d2.p_rotation = 1.0
    assert all(np.array_equal(d1._draw_matrix(2), d2._draw_matrix(2)) for _ in range(5))
    e = DeviceAugmenter((8, 8), (-1, 1), seed=4)
    none = sum(e._draw_matrix(2) is None for _ in range(4000)) / 4000
    assert abs(none - 0.64) < 0.03                      # (1 - 0.2)(1 - 0.2): neither rotation nor scaling drawn
