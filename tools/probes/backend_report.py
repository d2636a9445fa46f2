"""Which kernel family every dispatching module of the bench configurations took (nnuzoo_amd/backends.py), at 512^2.
Usage (GPU box): python tools/probes/backend_report.py"""
import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from nnuzoo_amd import backends as bk
from test_backends import _step
from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerM2Net, nnUNetTrainerSwT2Net
for cls in (nnUNetTrainerM2Net, nnUNetTrainerSwT2Net):
    tr = _step(cls, size=512)
    print(cls.__name__, bk.report(tr.network))
    from collections import Counter
    c = Counter((type(m).__name__, getattr(m, "in_features", None), getattr(m, "out_features", None))
                for m in tr.network.modules() if getattr(m, "backend", "hip").startswith("lib"))
    print("   library:", dict(c))
