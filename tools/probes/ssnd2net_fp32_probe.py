"""Why did SSND2NetP leave its parameters untouched over 24 fp32 steps (round 6, gpurun_out/r06_suite_a.log)?  One eager fp32 step:
which parameter gradients are non-finite, the gradient norm, whether the fused AdamW tail applied the update.
Usage (GPU box): python tools/probes/ssnd2net_fp32_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerSSND2NetP


class FP32(nnUNetTrainerSSND2NetP):
    _fp32_step = True
    _fp32_validation = True


def main():
    plans, cfg, dj = nnunet_plans(2, (128, 128), batch_size=2)
    torch.manual_seed(0)
    tr = FP32(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    tr.use_hip_graph = False
    b = synthetic_batch(2, (128, 128), tr._get_deep_supervision_scales(), seed=3)
    data, target = b["data"].cuda(), [t.cuda() for t in b["target"]]
    net = tr.network
    p0 = [p.detach().clone() for p in net.parameters()]
    out = net(data)
    l = tr.loss(list(out), target)
    print("loss", float(l), "max |logit|", [round(float(o.abs().max()), 2) for o in out])
    from nnuzoo_amd.token_linear import deferred_wgrads
    with deferred_wgrads():
        l.backward()
    bad = [(n, tuple(p.shape)) for n, p in net.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    none = [n for n, p in net.named_parameters() if p.grad is None]
    print("parameters", len(p0), "non-finite gradients", len(bad), bad[:8], "no gradient", len(none), none[:4])
    gn = torch.sqrt(sum(p.grad.double().pow(2).sum() for p in net.parameters() if p.grad is not None))
    print("gradient norm", float(gn))
    tr._optimizer_tail()
    moved = sum(int(not torch.equal(a, p.detach())) for a, p in zip(p0, net.parameters()))
    print("parameters changed by the tail", moved, "of", len(p0))


if __name__ == "__main__":
    main()
