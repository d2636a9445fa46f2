"""Round-4 fixtures for SegMamba's `MambaEncoder` (nets/seg_mamba/segmamba.py:158-222), generated in the BUILD CONTAINER from the
reference's own module under tools/ref_shim.py; only arrays are stored (tests/golden/), no reference source travels.
    python tools/make_golden_segmamba.py
The encoder is everything SegMamba defines itself: k7 s2 stem, InstanceNorm + k2 s2 down-samplings, GSC, MambaLayer (LayerNorm
-> the vendored bimamba "v3" (3-D) / "v2" (2-D) Mamba block of nets/seg_mamba/mamba_simple.py -> residual), InstanceNorm ->
MlpChannel per level.  Bindings: monai `Convolution(conv_only=True)` -> ref_shim.Convolution (nn.Sequential with a `conv` child);
the Mamba block's fused CUDA entry points -> the reference's own pure-torch definitions (`mamba_inner_ref`,
`selective_scan_ref`, exactly as tools/make_golden.py gen_mamba does).  The UNETR-style blocks around the encoder come from
monai (absent: unpinned) and are not part of this fixture.
Parameters: the reference's seeded construction + `InitWeights_He(1e-2)` (what get_seg_mamba_from_plans applies), stored in
the fixture in named_parameters() order.  Outputs: the four encoder feature maps, dx, and the L2 norm of every parameter
gradient for a fixed output-gradient pattern."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import ref_shim  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def pattern(shape, freq, phase):
    i = torch.arange(int(np.prod(shape)), dtype=torch.float64)
    return torch.cos(freq * i + phase).float().reshape(shape)


def main():
    torch.set_num_threads(4)
    ref = ref_shim.install()
    from nnunetv2.nets.seg_mamba import mamba_simple as ms
    inner_ref = ref_shim.load_mamba_inner_ref()
    ms.causal_conv1d_fn = None
    ms.selective_scan_fn = ref

    def no_out_proj(xz, cw, cb, xw, dw, A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None,
                    delta_softplus=True):
        eye = torch.eye(A.shape[0], dtype=xz.dtype)
        return inner_ref(xz, cw, cb, xw, dw, eye, None, A, B, C, D, delta_bias, B_proj_bias, C_proj_bias,
                         delta_softplus).transpose(1, 2)

    ms.mamba_inner_fn_no_out_proj = no_out_proj
    import nnunetv2.nets.seg_mamba.segmamba as R
    R.Convolution = ref_shim.Convolution
    from nnunetv2.utilities.network_initialization import InitWeights_He
    for tag, sd, shape, dims in (("3d", 3, (1, 1, 32, 32, 32), [16, 32, 48, 64]), ("2d", 2, (1, 2, 64, 64), [16, 32, 48, 64])):
        torch.manual_seed(0)
        enc = R.MambaEncoder(spatial_dims=sd, in_chans=shape[1], depths=[2, 2, 2, 2], dims=dims)
        enc.apply(InitWeights_He(1e-2))
        enc.eval()
        x = torch.randn(*shape, generator=torch.Generator().manual_seed(9))
        xg = x.clone().requires_grad_(True)
        outs = enc(xg)
        loss = 0
        for i, o in enumerate(outs):
            loss = loss + (o * pattern(o.shape, 0.37, 0.5 + i)).sum() / o[0, 0].numel()
        loss.backward()
        arr = {"x": x.numpy(), "dx": xg.grad.numpy(), "dims": np.array(dims)}
        for i, o in enumerate(outs):
            arr[f"out{i}"] = o.detach().numpy()
        names, norms = [], []
        for n, p in enc.named_parameters():
            arr[f"p_{n}"] = p.detach().numpy()
            if p.grad is not None:
                names.append(n)
                norms.append(float(p.grad.double().pow(2).sum().sqrt()))
        arr["grad_names"] = np.array(names)
        arr["grad_norms"] = np.array(norms)
        # conditioning of the reference itself: relative output change for a 1e-6 relative input perturbation
        with torch.no_grad():
            o2 = enc(x + 1e-6 * float(x.std()) * pattern(x.shape, 1.3, 0.2))
        arr["sens"] = np.array([float((a - b).abs().max() / b.abs().max()) for a, b in zip(o2, outs)])
        path = os.path.join(OUT, f"segmamba_encoder_{tag}.npz")
        np.savez_compressed(path, **arr)
        print(tag, "outs", [tuple(o.shape) for o in outs], "sens", arr["sens"], "params", len(list(enc.parameters())),
              round(os.path.getsize(path) / 2 ** 20, 2), "MB", flush=True)


if __name__ == "__main__":
    main()
