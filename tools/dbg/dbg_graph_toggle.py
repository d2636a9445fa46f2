"""Diagnostic (GPU box): which fused op breaks the hipGraph replay.
Usage: python tools/dbg/dbg_graph_toggle.py noln,noxs,nofin [bench_zoo args]"""
import os
import sys

import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
off = set(sys.argv[1].split(","))
from nnuzoo_amd import layer_norm as LN
from nnuzoo_amd.nets.m2net import SS2D
from nnuzoo_amd.training.loss import DC_and_CE_loss
if "noln" in off:
    LN.LayerNorm.forward = lambda self, x: F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
if "noxs" in off:
    SS2D.fused_cross_scan = False
if "nofin" in off:
    DC_and_CE_loss.finalize_on_device = lambda self: False
sys.argv = [os.path.join(ROOT, "tools", "bench_zoo.py")] + sys.argv[2:]
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_zoo
bench_zoo.main()
