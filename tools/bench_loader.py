"""Throughput of the device-resident loader (nnuzoo_amd/dataloading/device_loader.py) at the bench configuration: batch 2 of
1x128^3 patches with the 5 deep-supervision targets, cut from synthetic resident cases.  Usage (GPU box):
python tools/bench_loader.py [--cases 8] [--edge 128]"""
import argparse
import os
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnuzoo_amd.dataloading.device_loader import DeviceCaseStore, nnUNetDataLoader


class _Cases:
    def __init__(self, n, shape):
        rs = np.random.RandomState(0)
        self.identifiers = [f"c{i}" for i in range(n)]
        self.c = {}
        for k in self.identifiers:
            seg = (rs.rand(1, *shape) > 0.97).astype(np.int16)
            idx = np.argwhere(seg == 1)
            self.c[k] = (rs.randn(1, *shape).astype(np.float32), seg, None,
                         {"class_locations": {1: idx[rs.choice(len(idx), 10000)]}})

    def load_case(self, k):
        return self.c[k]


ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=8)
ap.add_argument("--edge", type=int, default=128)
a = ap.parse_args()
e = a.edge
store = DeviceCaseStore(_Cases(a.cases, (e + 40, e + 72, e + 72)))
scales = [[1 / 2 ** i] * 3 for i in range(5)]
lm = types.SimpleNamespace(all_labels=[0, 1], has_ignore_label=False)
dl = nnUNetDataLoader(store, 2, (e, e, e), (e, e, e), lm, oversample_foreground_percent=0.33, deep_supervision_scales=scales,
                      mirror_axes=(0, 1, 2))
for _ in range(5):
    dl.generate_train_batch()
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
s, f = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(n):
    b = dl.generate_train_batch()
f.record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n
gpu = s.elapsed_time(f) / n
byt = 2 * e ** 3 * (4 + 2) * 2 + sum(2 * 2 * int(round(e * sc[0])) ** 3 * 2 for sc in scales[1:])   # read + write
print(f"resident cases {store.nbytes() / 1e9:.2f} GB; batch of 2x1x{e}^3 + {len(scales)} targets: host {wall * 1e3:.3f} ms, "
      f"GPU {gpu * 1e3:.1f} us per batch -> {2 / wall:.0f} patches/s from one process ({byt / gpu / 1e6:.0f} GB/s moved)")
