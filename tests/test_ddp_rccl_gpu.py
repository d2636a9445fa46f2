"""The data-parallel code paths on the real backend ("nccl" = RCCL) at world size 1, in a child process: the 3-D nnUNet
trainer with the arena all-reduce driven by the backward schedule, and a zoo trainer (SyncBatchNorm conversion, rank-0
broadcast, batch-Dice AllGatherGrad, gradient averaging after backward).  World size > 1 is covered on CPU with gloo
(tests/test_ddp_gloo.py); the 8-GPU curve is the driver's run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, json
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
from nnuzoo_amd.training.nnUNetTrainer import nnUNetTrainer
from nnuzoo_amd.training import zoo_trainers as Z
out = {}
plans, cfg, dj = nnunet_plans(3, (32, 32, 32), batch_size=2)
torch.manual_seed(0)
tr = nnUNetTrainer(plans, cfg, 0, dj, device=torch.device("cuda"))
tr.initialize()
assert tr.is_ddp and hasattr(tr.network, "grad_reducer")
b = synthetic_batch(2, (32, 32, 32), tr._get_deep_supervision_scales(), seed=1)
out["unet"] = [float(tr.train_step(b)["loss"]) for _ in range(3)]
# round 4: the data-parallel step as hipGraph segments with the collectives between them - against the eager DDP step
def run(graph):
    os.environ["NNZ_DDP_GRAPH"] = "1" if graph else "0"
    plans, cfg, dj = nnunet_plans(3, (32, 32, 32), batch_size=2)
    torch.manual_seed(0)
    t = nnUNetTrainer(plans, cfg, 0, dj, device=torch.device("cuda"))
    t.initialize()
    ls = []
    for i in range(4):
        bb = synthetic_batch(2, (32, 32, 32), t._get_deep_supervision_scales(), seed=40 + i)
        ls.append(float(t.train_step(bb)["loss"]))
    return ls, [p.detach().clone() for p in t.network.parameters()], t
le, pe, te = run(False)
lg, pg, tg = run(True)
assert te._graphed_ddp is None and tg._graphed_ddp is not None
out["ddp_graph"] = {"segments": len(tg._graphed_ddp.segments), "buckets": tg.network.grad_reducer.buckets_last_step,
                    "eager_buckets": te.network.grad_reducer.buckets_last_step, "losses_equal": le == lg,
                    "params_equal": all(torch.equal(a, b) for a, b in zip(pe, pg)), "losses": lg}
plans, cfg, dj = nnunet_plans(2, (64, 64), batch_size=2)
torch.manual_seed(0)
zt = Z.nnUNetTrainerM2NetP(plans, cfg, 0, dj, device=torch.device("cuda"))
zt.initialize()
assert zt.is_ddp
assert any(isinstance(m, torch.nn.SyncBatchNorm) for m in zt.network.modules())
assert not any(type(m) is torch.nn.BatchNorm2d for m in zt.network.modules())
b = synthetic_batch(2, (64, 64), zt._get_deep_supervision_scales(), seed=2)
out["zoo"] = [float(zt.train_step(b)["loss"]) for _ in range(3)]
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
"""


def test_ddp_paths_on_rccl_world1(hip_lib):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, timeout=600, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert r.returncode == 0 and lines, r.stderr[-3000:]
    import json
    import math
    res = json.loads(lines[-1][7:])
    assert all(math.isfinite(v) for v in res["unet"] + res["zoo"]), res
    # graph segments = buckets (+1 when the last segment carries no slice); at world size 1 the averaged and the summed
    # gradients are the same numbers, so the segmented replay must equal the eager DDP step bit for bit
    g = res["ddp_graph"]
    assert g["segments"] >= 2 and g["buckets"] == g["eager_buckets"] >= 1, g
    assert g["losses_equal"] and g["params_equal"], g
