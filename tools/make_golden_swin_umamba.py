"""Round-4 fixtures for Swin-UMamba / Swin-UMamba-D, generated in the BUILD CONTAINER from the reference's own modules
(/root/reference/nnunetv2/nets/SwinUMamba.py, SwinUMambaD.py) under tools/ref_shim.py; only arrays, names, shapes and digests are
stored (tests/golden/), no reference source travels.
    python tools/make_golden_swin_umamba.py
1. tests/golden/swin_umamba_manifest.json
     * `SwinUMambaD`: state_dict names + shapes of get_swin_umamba_d_from_plans' network (1 input channel, 3 heads) and the digest
       of its parameters after torch.manual_seed(0) + the factory (construction order, the RNG-advancing no-op init of VSSLayer,
       InitWeights_He);
     * `SwinUMamba.vssm_encoder`: the same for VSSMEncoder(patch_size=2, in_chans=48) - the part of Swin-UMamba the reference
       defines itself (the UNETR blocks around it are monai's, absent here).
2. tests/golden/swin_umamba_d.npz: a reduced SwinUMambaD (dims 16..128, depths [1, 1, 2, 1], 2 input channels, 3 heads, deep
   supervision) with the reference's parameters, input, the four outputs, dx, the L2 norm of every parameter gradient, and the
   reference's own response to a 1e-6 input perturbation.  selective_scan_fn is bound to the reference's selective_scan_ref."""
import hashlib
import json
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import ref_shim  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def pattern(shape, freq, phase):
    i = torch.arange(int(np.prod(shape)), dtype=torch.float64)
    return torch.cos(freq * i + phase).float().reshape(shape)


def state_digest(sd):
    h = hashlib.sha256()
    crc = {}
    for i, (k, v) in enumerate(sd.items()):
        b = v.detach().cpu().contiguous().numpy().tobytes()
        h.update(b)
        if i % 40 == 0:
            crc[k] = zlib.crc32(b)
    return {"sha256": h.hexdigest(), "n_tensors": len(sd), "crc32": crc}


def main():
    torch.set_num_threads(4)
    ref = ref_shim.install()
    import nnunetv2.nets.SwinUMamba as RS
    import nnunetv2.nets.SwinUMambaD as RD
    RS.selective_scan_fn = ref
    RD.selective_scan_fn = ref

    class _LM:
        num_segmentation_heads = 3

    class _PM:
        def get_label_manager(self, dataset_json):
            return _LM()

    class _CM:
        conv_kernel_sizes = [[3, 3]]

    manifest = {}
    torch.manual_seed(0)
    net = RD.get_swin_umamba_d_from_plans(_PM(), {}, _CM(), 1, deep_supervision=True, use_pretrain=False)
    manifest["SwinUMambaD"] = {"state_dict": [[k, list(v.shape)] for k, v in net.state_dict().items()],
                               "seeded": state_digest(net.state_dict())}
    torch.manual_seed(0)
    enc = RS.VSSMEncoder(patch_size=2, in_chans=48)
    manifest["SwinUMamba.vssm_encoder"] = {"state_dict": [[k, list(v.shape)] for k, v in enc.state_dict().items()],
                                           "seeded": state_digest(enc.state_dict())}
    json.dump(manifest, open(os.path.join(OUT, "swin_umamba_manifest.json"), "w"))
    print("manifest", {k: len(v["state_dict"]) for k, v in manifest.items()}, flush=True)

    dims = [16, 32, 64, 128]
    torch.manual_seed(1)
    net = RD.SwinUMambaD(dict(in_chans=2, patch_size=4, depths=[1, 1, 2, 1], dims=dims, drop_path_rate=0.2),
                         dict(num_classes=3, deep_supervision=True, features_per_stage=dims, drop_path_rate=0.2, d_state=16))
    from nnunetv2.utilities.network_initialization import InitWeights_He
    net.apply(InitWeights_He(1e-2))
    with torch.no_grad():          # the factory's zero biases / unit norms carry no information: give every parameter a value
        g = torch.Generator().manual_seed(5)
        for n, p in net.named_parameters():
            if p.ndim == 1 and ("norm" in n or "ln_1" in n or n.endswith(".bias")) and "dt_projs" not in n:
                p.add_(0.05 * torch.randn(p.shape, generator=g))
    net.eval()
    x = torch.randn(1, 2, 64, 64, generator=torch.Generator().manual_seed(9))
    xg = x.clone().requires_grad_(True)
    outs = net(xg)
    loss = 0
    for i, o in enumerate(outs):
        loss = loss + (o * pattern(o.shape, 0.37, 0.5 + i)).sum() / o[0, 0].numel()
    loss.backward()
    arr = {"x": x.numpy(), "dx": xg.grad.numpy(), "dims": np.array(dims)}
    for i, o in enumerate(outs):
        arr[f"out{i}"] = o.detach().numpy()
    names, norms = [], []
    for n, p in net.named_parameters():
        arr[f"p_{n}"] = p.detach().numpy()
        if p.grad is not None:
            names.append(n)
            norms.append(float(p.grad.double().pow(2).sum().sqrt()))
    arr["grad_names"] = np.array(names)
    arr["grad_norms"] = np.array(norms)
    with torch.no_grad():
        o2 = net(x + 1e-6 * float(x.std()) * pattern(x.shape, 1.3, 0.2))
    arr["sens"] = np.array([float((a - b).abs().max() / b.abs().max()) for a, b in zip(o2, outs)])
    path = os.path.join(OUT, "swin_umamba_d.npz")
    np.savez_compressed(path, **arr)
    print("outs", [tuple(o.shape) for o in outs], "sens", arr["sens"], "params", len(names),
          round(os.path.getsize(path) / 2 ** 20, 2), "MB", flush=True)


if __name__ == "__main__":
    main()
