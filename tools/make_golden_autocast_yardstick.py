"""Records the yardstick of tests/test_plain_unet_gpu.py::test_full_size_3d_fullres_128 on an MI355X: the relative error of
every parameter gradient of torch's own fp16-autocast step (MIOpen / ATen kernels - the numerics of the reference trainer,
nnUNetTrainer.py:1012-1022) against the fp32 CPU oracle, on the test's own network, patch and output gradients.

    gpurun -- 'python tools/make_golden_autocast_yardstick.py gpurun_out/plainconv_128_autocast_yardstick.json'

then copy the file to tests/golden/.  ~4 min (MIOpen probes its solvers at 128^3).  No product code runs here: oracle
(CPU, fp32) against a CUDA copy of the oracle under torch.autocast.
"""
import copy
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from tests.test_plain_unet_gpu import full_size_case  # noqa: E402


def main():
    out = sys.argv[1]
    case, ref, _net, x, g = full_size_case()
    outs_ref = ref(x)
    gs = [torch.randn(r.shape, generator=g).to(torch.float16).float() for r in outs_ref]
    wts = [1.0, 0.5, 0.25, 0.125, 0.0625][: len(outs_ref)]
    wts[-1] = 0.0
    sum(w * (o * G).sum() for w, o, G in zip(wts, outs_ref, gs) if w != 0).backward()
    ref16 = copy.deepcopy(ref).cuda()
    for p in ref16.parameters():
        p.grad = None
    with torch.autocast("cuda", dtype=torch.float16):
        o16 = ref16(x.cuda())
    sum(w * (o.float() * G.cuda()).sum() for w, o, G in zip(wts, o16, gs) if w != 0).backward()
    torch.cuda.synchronize()
    rel = {}
    ac = dict(ref16.named_parameters())
    for name, p in ref.named_parameters():
        if p.grad is None:
            continue
        rel[name] = float((ac[name].grad.float().cpu() - p.grad).norm() / (p.grad.norm() + 1e-12))
    doc = {"case": "3d_fullres_128, 1x128^3, seeds of tests/test_plain_unet_gpu.py::full_size_case",
           "device": torch.cuda.get_device_name(0), "torch": torch.__version__,
           "what": "||grad(fp16 autocast, torch kernels) - grad(fp32 CPU oracle)|| / ||grad(fp32 CPU oracle)|| per parameter",
           "rel16": rel}
    json.dump(doc, open(out, "w"), indent=1)
    print("wrote", out, len(rel), "parameters, max", max(rel.values()))


if __name__ == "__main__":
    main()
