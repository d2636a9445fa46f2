"""Device-side training augmentations (csrc/augment.hip, nnuzoo_amd/dataloading/device_augment.py): every launch
against a plain torch fp32 formulation of the same arithmetic, through the C-ABI.  The transform classes of the reference's chain
(nnUNetTrainer.py:825-973) live in batchgeneratorsv2, which is absent here: parity with THAT package is unpinned;
what is held is that each kernel computes the arithmetic its header states."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from nnuzoo_amd import _lib
from nnuzoo_amd._lib import call, ptr, stream_ptr

DEV = "cuda"


def _grid_sample_reference(x, m):
    """x (B, C, D, H, W) on the CPU in float64-free plain torch: sample at M (o - c) + c with zeros outside (align_corners=True
    normalised coordinates are index coordinates)"""
    B, C, D, H, W = x.shape
    out = torch.empty_like(x)
    zz, yy, xx = torch.meshgrid(torch.arange(D, dtype=torch.float32), torch.arange(H, dtype=torch.float32),
                                torch.arange(W, dtype=torch.float32), indexing="ij")
    c = torch.tensor([(D - 1) / 2, (H - 1) / 2, (W - 1) / 2])
    o = torch.stack([zz, yy, xx], -1) - c
    for b in range(B):
        src = o @ torch.tensor(m[b], dtype=torch.float32).t() + c                    # (D, H, W, 3) source index (z, y, x)

        def norm(v, n):
            return 2 * v / (n - 1) - 1 if n > 1 else torch.zeros_like(v)
        g = torch.stack([norm(src[..., 2], W), norm(src[..., 1], H), norm(src[..., 0], D)], -1)[None]
        out[b] = F.grid_sample(x[b:b + 1], g, mode="bilinear", padding_mode="zeros", align_corners=True)[0]
    return out


def _rot(ax, t):
    c, s = math.cos(t), math.sin(t)
    m = np.eye(3)
    i, j = [(1, 2), (0, 2), (0, 1)][ax]
    m[i, i], m[i, j], m[j, i], m[j, j] = c, -s, s, c
    return m


@pytest.mark.parametrize("shape", [(2, 2, 12, 20, 16), (3, 1, 1, 24, 24)])
def test_affine_resampling_matches_grid_sample_and_nearest(hip_lib, shape):
    B, C, D, H, W = shape
    g = torch.Generator().manual_seed(7)
    x = torch.randn(shape, generator=g)
    seg = torch.randint(-1, 4, (B, 1, D, H, W), generator=g).to(torch.int16)
    mats = []
    for b in range(B):
        m = _rot(0, 0.3 + 0.2 * b) * (0.8 + 0.3 * b)
        if D > 1:
            m = _rot(1, -0.2) @ _rot(2, 0.15) @ m
        mats.append(m)
    flat = np.zeros((B, 12), dtype=np.float32)
    for b in range(B):
        flat[b].reshape(3, 4)[:, :3] = mats[b]
    xd, out = x.to(DEV), torch.empty(shape, device=DEV)
    call("nnz_aug_affine_f32", ptr(xd), ptr(out), flat.ctypes.data, B, C, D, H, W, 0.0, stream_ptr())
    ref = _grid_sample_reference(x, [m.astype(np.float32) for m in mats])
    assert (out.cpu() - ref).abs().max().item() <= 2e-5 * x.abs().max().item()
    sd, so = seg.to(DEV), torch.empty(seg.shape, dtype=torch.int16, device=DEV)
    call("nnz_aug_affine_i16", ptr(sd), ptr(so), flat.ctypes.data, B, 1, D, H, W, -1, stream_ptr())
    # nearest: floor(s + 0.5) per axis, -1 outside
    zz, yy, xx = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
    c = np.array([(D - 1) / 2, (H - 1) / 2, (W - 1) / 2], dtype=np.float32)
    o = np.stack([zz, yy, xx], -1).astype(np.float32) - c
    got = so.cpu().numpy()
    mism = 0
    for b in range(B):
        src = o @ mats[b].astype(np.float32).T + c
        idx = np.floor(src + 0.5).astype(np.int64)
        if D == 1:
            idx[..., 0] = 0
        ok = (idx[..., 0] >= 0) & (idx[..., 0] < D) & (idx[..., 1] >= 0) & (idx[..., 1] < H) & (idx[..., 2] >= 0) & (idx[..., 2] < W)
        want = np.full((D, H, W), -1, dtype=np.int16)
        ii = np.clip(idx, 0, [D - 1, H - 1, W - 1])
        want[ok] = seg[b, 0].numpy()[ii[..., 0], ii[..., 1], ii[..., 2]][ok]
        # source coordinates within 1e-4 of a rounding boundary may fall either way (fp32 matrix product order)
        frac = np.abs((src + 0.5) - np.round(src + 0.5)).min(-1)
        mism += int(((got[b, 0] != want) & (frac > 1e-4)).sum())
    assert mism == 0


def test_identity_is_a_copy_and_quarter_turn_is_rot90(hip_lib):
    B, C, D, H, W = 2, 3, 1, 32, 32
    x = torch.randn(B, C, D, H, W, device=DEV)
    flat = np.zeros((B, 12), dtype=np.float32)
    flat[0].reshape(3, 4)[:, :3] = np.eye(3)
    flat[1].reshape(3, 4)[:, :3] = np.round(_rot(0, math.pi / 2))            # exact 0 / +-1 entries
    out = torch.empty_like(x)
    call("nnz_aug_affine_f32", ptr(x), ptr(out), flat.ctypes.data, B, C, D, H, W, 0.0, stream_ptr())
    assert torch.equal(out[0], x[0])
    # source (y, x) = (-(ox - c) + c, (oy - c) + c): a quarter turn of the image
    assert torch.equal(out[1, :, 0], torch.rot90(x[1, :, 0], k=-1, dims=(1, 2))) or \
        torch.equal(out[1, :, 0], torch.rot90(x[1, :, 0], k=1, dims=(1, 2)))


def test_statistics_and_intensity_transforms(hip_lib):
    nbc, n = 6, 50_000
    g = torch.Generator().manual_seed(3)
    x0 = (torch.randn(nbc, n, generator=g) * torch.linspace(0.5, 3, nbc)[:, None] + torch.linspace(-2, 2, nbc)[:, None])
    x = x0.to(DEV)
    lib = _lib.load()
    ws = torch.empty(int(lib.nnz_aug_stats_workspace_floats(nbc)), device=DEV)
    st = torch.empty(nbc, 4, device=DEV)
    call("nnz_aug_stats_f32", ptr(x), n, nbc, ptr(ws), ptr(st), stream_ptr())
    s = st.cpu()
    d = x0.double()
    assert torch.allclose(s[:, 0].double(), d.mean(1), atol=1e-6) and torch.allclose(s[:, 1].double(), d.std(1, unbiased=False), rtol=1e-5)
    assert torch.equal(s[:, 2], x0.min(1).values) and torch.equal(s[:, 3], x0.max(1).values)
    st2 = torch.empty(nbc, 4, device=DEV)
    call("nnz_aug_stats_f32", ptr(x), n, nbc, ptr(ws), ptr(st2), stream_ptr())
    assert torch.equal(st, st2)                                            # fixed summation order
    rec = torch.zeros(nbc, 4)
    rec[:, 0] = torch.tensor([1, 1, 0, 1, 1, 0.])                          # rows 2 and 5 stay untouched
    rec[:, 1] = torch.tensor([0.8, 1.2, 9, 0.75, 1.4, 9])
    rec[:, 2] = torch.tensor([0.1, -0.2, 9, 0, 0, 9])
    r = rec.to(DEV)
    on = rec[:, 0] > 0

    def run(op, sa=None, sb=None, seed=0):
        y = x.clone()
        call("nnz_aug_intensity_f32", ptr(y), n, nbc, op, ptr(r), ptr(sa), ptr(sb), seed, stream_ptr())
        return y.cpu()
    # linear
    y = run(1)
    want = torch.where(on[:, None], rec[:, 1:2] * x0 + rec[:, 2:3], x0)
    assert torch.allclose(y, want, rtol=1e-6, atol=1e-6) and torch.equal(y[~on], x0[~on])
    # contrast with range preservation
    y = run(2, sa=st)
    want = torch.where(on[:, None], torch.minimum(torch.maximum((x0 - s[:, 0:1]) * rec[:, 1:2] + s[:, 0:1], s[:, 2:3]), s[:, 3:4]), x0)
    assert torch.allclose(y, want, rtol=1e-5, atol=1e-5)
    # gamma on the [min, max] range
    y = run(3, sa=st)
    rng = (s[:, 3:4] - s[:, 2:3])
    want = torch.where(on[:, None], ((x0 - s[:, 2:3]) / rng.clamp_min(1e-7)).clamp_min(0).pow(rec[:, 1:2]) * rng + s[:, 2:3], x0)
    assert torch.allclose(y, want, rtol=2e-5, atol=2e-5)
    # restore mean / std: after a gamma the statistics of the active rows come back to `st`
    yg = y.to(DEV)
    st_after = torch.empty(nbc, 4, device=DEV)
    call("nnz_aug_stats_f32", ptr(yg), n, nbc, ptr(ws), ptr(st_after), stream_ptr())
    call("nnz_aug_intensity_f32", ptr(yg), n, nbc, 4, ptr(r), ptr(st_after), ptr(st), 0, stream_ptr())
    back = yg.cpu().double()
    assert torch.allclose(back.mean(1)[on], d.mean(1)[on], atol=1e-4) and torch.allclose(back.std(1, unbiased=False)[on], d.std(1, unbiased=False)[on], rtol=1e-4)
    # noise: zero mean, the requested sigma, reproducible from the seed, different per row and per seed
    rec[:, 1] = torch.tensor([0.1, 0.3, 9, 0.2, 0.05, 9])
    r = rec.to(DEV)
    a, b2, c = run(0, seed=123), run(0, seed=123), run(0, seed=124)
    assert torch.equal(a, b2) and not torch.equal(a, c)
    noise = (a - x0)[on]
    assert noise.mean(1).abs().max() < 4 * 0.3 / math.sqrt(n)
    assert torch.allclose(noise.std(1), rec[on, 1], rtol=0.03)
    assert (noise[0, :1000] - noise[1, :1000] / 3).abs().max() > 1e-3         # rows draw their own stream
    assert torch.equal(a[~on], x0[~on])


def test_blur_and_low_resolution(hip_lib):
    nbc, D, H, W = 4, 10, 14, 12
    g = torch.Generator().manual_seed(9)
    x0 = torch.randn(nbc, D, H, W, generator=g)
    x = x0.to(DEV)
    rec = torch.tensor([[1, 0.6, 0.9, 0.75], [0, 1, 1, 1], [1, 0.5, 0.0, 1.0], [1, 1.0, 1.0, 1.0]])
    r = rec.to(DEV)
    cur = x
    for axis in range(3):
        out = torch.empty_like(cur)
        call("nnz_aug_blur_axis_f32", ptr(cur), ptr(out), nbc, D, H, W, axis, ptr(r), stream_ptr())
        cur = out
    got = cur.cpu()
    want = x0.clone()
    for bc in range(nbc):
        if rec[bc, 0] == 0:
            continue
        v = x0[bc]
        for axis in range(3):
            sg = float(rec[bc, 1 + axis])
            if sg <= 0:
                continue
            R = min(4, max(1, math.ceil(3 * sg)))
            k = torch.arange(-R, R + 1, dtype=torch.float32)
            w = torch.exp(-k * k / (2 * sg * sg))
            w = w / w.sum()
            n = v.shape[axis]
            idx = (torch.arange(n)[:, None] + k.long()[None]).clamp(0, n - 1)          # (n, taps): edge voxels repeated
            v = (v.movedim(axis, -1)[..., idx] * w).sum(-1).movedim(-1, axis)
        want[bc] = v
    assert torch.equal(got[1], x0[1])
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5)
    # low resolution: nearest down to round(n * scale), linear up (positions (j + 0.5) n / m - 0.5, clamped)
    rec = torch.tensor([[1, 0.5, 0, 0], [0, 0.7, 0, 0], [1, 0.83, 0, 0], [1, 1.0, 0, 0]])
    r = rec.to(DEV)
    out = torch.empty_like(x)
    call("nnz_aug_lowres_f32", ptr(x), ptr(out), nbc, D, H, W, 0, ptr(r), stream_ptr())
    got = out.cpu()
    assert torch.equal(got[1], x0[1]) and torch.equal(got[3], x0[3])               # inactive row; scale 1 = nothing to do

    def axis_maps(n, sc):
        m = max(1, int(np.rint(np.float32(n) * np.float32(sc))))
        j = np.arange(m, dtype=np.float32)
        src = np.minimum(np.floor((j + np.float32(0.5)) * np.float32(n) / np.float32(m)).astype(np.int64), n - 1)
        i = np.arange(n, dtype=np.float32)
        p = np.clip((i + np.float32(0.5)) * np.float32(m) / np.float32(n) - np.float32(0.5), 0, m - 1).astype(np.float32)
        j0 = np.floor(p).astype(np.int64)
        j1 = np.minimum(j0 + 1, m - 1)
        return src, j0, j1, torch.from_numpy((p - j0).astype(np.float32))
    for bc in (0, 2):
        sc = float(rec[bc, 1])
        v = x0[bc]
        for axis, n in enumerate((D, H, W)):
            src, j0, j1, t = axis_maps(n, sc)
            low = v.movedim(axis, -1)[..., src]
            v = (low[..., j0] * (1 - t) + low[..., j1] * t).movedim(-1, axis)
        assert torch.allclose(got[bc], v, rtol=1e-5, atol=1e-5)
    out2 = torch.empty_like(x)
    call("nnz_aug_lowres_f32", ptr(x), ptr(out2), nbc, D, H, W, 1, ptr(r), stream_ptr())      # keep_z: planes stay separate
    v = x0[0]
    for axis, n in ((1, H), (2, W)):
        src, j0, j1, t = axis_maps(n, 0.5)
        low = v.movedim(axis, -1)[..., src]
        v = (low[..., j0] * (1 - t) + low[..., j1] * t).movedim(-1, axis)
    assert torch.allclose(out2.cpu()[0], v, rtol=1e-5, atol=1e-5)


def test_augmenter_chain_and_loader_hook(hip_lib):
    from nnuzoo_amd.dataloading.device_augment import DeviceAugmenter
    aug = DeviceAugmenter((32, 32, 32), (-0.5, 0.5), seed=11)
    aug.p_rotation = aug.p_scaling = aug.p_noise = aug.p_brightness = aug.p_contrast = aug.p_gamma = aug.p_gamma_inverted = 1.0
    aug.p_blur = aug.p_blur_per_channel = aug.p_lowres = aug.p_lowres_per_channel = 1.0
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 2, 32, 32, 32, generator=g).to(DEV)
    seg = torch.randint(-1, 3, (2, 1, 32, 32, 32), generator=g).to(torch.int16).to(DEV)
    y, s = aug(x.clone(), seg.clone())
    assert y.shape == x.shape and s.shape == seg.shape and torch.isfinite(y).all()
    assert int((s == -1).sum()) == 0 and set(torch.unique(s).tolist()) <= {0, 1, 2}
    assert all(m is not None for m in aug.last["matrices"]) and np.isfinite(aug.last["gamma"]).all()
    assert np.isfinite(aug.last["blur_sigma"]).all() and np.isfinite(aug.last["lowres_scale"]).all()
    # nothing drawn: data untouched, only the -1 label is rewritten
    aug0 = DeviceAugmenter((32, 32, 32), (-0.5, 0.5), seed=1)
    aug0.p_rotation = aug0.p_scaling = aug0.p_noise = aug0.p_brightness = aug0.p_contrast = aug0.p_gamma = aug0.p_gamma_inverted = 0.0
    aug0.p_blur = aug0.p_lowres = 0.0
    y0, s0 = aug0(x.clone(), seg.clone())
    assert torch.equal(y0, x) and torch.equal(s0, torch.where(seg == -1, torch.zeros_like(seg), seg))
    # same seed, same batch -> same result; default probabilities over many samples hit the call site's rates
    a1, a2 = DeviceAugmenter((32, 32, 32), (-0.5, 0.5), seed=4), DeviceAugmenter((32, 32, 32), (-0.5, 0.5), seed=4)
    assert torch.equal(a1(x.clone(), None)[0], a2(x.clone(), None)[0])
    rate = DeviceAugmenter((8, 8), (-3.14, 3.14), seed=2)
    xs = torch.randn(16, 1, 8, 8, device=DEV)
    hits = {"rot_or_scale": 0, "gamma": 0, "noise": 0}
    for _ in range(60):
        rate(xs.clone(), None)
        hits["rot_or_scale"] += sum(m is not None for m in rate.last["matrices"])
        hits["gamma"] += int(np.isfinite(rate.last["gamma"][:, 0]).sum())
        hits["noise"] += sum(v is not None for v in rate.last["noise_sigma"])
    n = 60 * 16
    assert abs(hits["rot_or_scale"] / n - 0.36) < 0.06 and abs(hits["gamma"] / n - 0.3) < 0.06 and abs(hits["noise"] / n - 0.1) < 0.04
