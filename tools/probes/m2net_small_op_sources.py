"""Where do the ~1 500 small ATen launches (copies, fills, casts, adds) of an M2Net step come from?  One eager step under
torch.profiler with Python stacks, grouped by (op, innermost nnuzoo_amd frame).
Usage (GPU box): python tools/probes/m2net_small_op_sources.py [M2Net]"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["NNZ_HIP_GRAPH"] = "0"
from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
from nnuzoo_amd.training import zoo_trainers as Z

name = sys.argv[1] if len(sys.argv) > 1 else "M2Net"
plans, cfg, dj = nnunet_plans(2, (512, 512), batch_size=2)
torch.manual_seed(0)
tr = getattr(Z, "nnUNetTrainer" + name)(plans, cfg, 0, dj, device=torch.device("cuda"))
tr.initialize()
b = synthetic_batch(2, (512, 512), tr._get_deep_supervision_scales(), seed=3)
b = {"data": b["data"].cuda(), "target": [t.cuda() for t in b["target"]]}
for _ in range(3):
    tr.train_step(b)
torch.cuda.synchronize()
import traceback

from torch.utils._python_dispatch import TorchDispatchMode

WATCH = ("copy_", "fill_", "zero_", "add", "add_", "_to_copy", "cat", "mul", "clone", "sum", "zeros", "zeros_like", "div",
         "sub", "neg", "where", "sigmoid", "silu", "gelu", "permute_copy", "native_dropout", "mean", "expand_copy")
cnt = collections.Counter()


class Tap(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in WATCH:
            site = "(autograd engine: no python frame of the package)"
            for fr in reversed(traceback.extract_stack(limit=40)):
                if "nnuzoo_amd/" in fr.filename and "probes" not in fr.filename:
                    site = f"{fr.filename.split('nnuzoo_amd/')[-1]}:{fr.lineno} {fr.name}"
                    break
            if os.environ.get("NNZ_PROBE_SHAPES") == "1" and site.startswith("(autograd"):
                t = next((a for a in args if torch.is_tensor(a)), None)
                if t is None and args and isinstance(args[0], (list, tuple)) and args[0] and torch.is_tensor(args[0][0]):
                    t = args[0][0]
                if t is not None:
                    site = f"engine {str(t.dtype).replace('torch.', '')} {tuple(t.shape)} contiguous={t.is_contiguous()}"
            cnt[(name, site)] += 1
        return func(*args, **(kwargs or {}))


with Tap():
    tr.train_step(b)
    torch.cuda.synchronize()
for (op, site), n in cnt.most_common(70 if os.environ.get("NNZ_PROBE_SHAPES") != "1" else 200):
    print(f"{n:5d}  {op:14s} {site}")
