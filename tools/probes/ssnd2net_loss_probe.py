"""Why does the SSND2Net 512^2 bench loss stay flat?  Logs per eager step: loss, GradScaler scale, whether
the step was skipped (non-finite gradients), the pre-clip gradient norm and the largest |logit| per output.
Usage (GPU box): python tools/probes/ssnd2net_loss_probe.py [--size 512] [--steps 12] [--model SSND2Net] [--fp32 0]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
from nnuzoo_amd.training import zoo_trainers as Z


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="SSND2Net")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--fp32", type=int, default=0)
    ap.add_argument("--eval-droppath", type=int, default=0)
    a = ap.parse_args()
    cls = getattr(Z, "nnUNetTrainer" + a.model)
    plans, cfg, dj = nnunet_plans(2, (a.size, a.size), batch_size=2)
    torch.manual_seed(0)
    tr = cls(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    tr.use_hip_graph = False
    net = tr.network
    b = synthetic_batch(2, (a.size, a.size), tr._get_deep_supervision_scales(), seed=3)
    data = b["data"].cuda()
    target = [t.cuda() for t in b["target"]]
    for step in range(a.steps):
        tr.optimizer.zero_grad(set_to_none=True)
        ctx = torch.autocast('cuda', enabled=not a.fp32)
        with ctx:
            out = net(data)
            l = tr.loss(list(out), target)
        rec = {"step": step, "loss": round(float(l), 5), "scale": tr.grad_scaler.get_scale()}
        rec["max_abs_logit"] = [round(float(o.float().abs().max()), 2) for o in out]
        rec["nonfinite_logit"] = [int((~torch.isfinite(o)).sum()) for o in out]
        if a.fp32:
            l.backward()
        else:
            tr.grad_scaler.scale(l).backward()
            tr.grad_scaler.unscale_(tr.optimizer)
        bad = [(n, tuple(p.shape)) for n, p in net.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        rec["params_with_nonfinite_grad"] = len(bad)
        rec["first_bad"] = [n for n, _ in bad[:6]]
        rec["last_bad"] = [n for n, _ in bad[-3:]]
        gn = torch.nn.utils.clip_grad_norm_(net.parameters(), 12)
        rec["grad_norm"] = float(gn)
        if a.fp32:
            tr.optimizer.step()
        else:
            tr.grad_scaler.step(tr.optimizer)
            tr.grad_scaler.update()
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
