"""Probe (GPU box): which autograd op of the SwT2Net step launches the 20 ms library batched-GEMM weight-gradient kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
from nnuzoo_amd.training import zoo_trainers as Z
os.environ["NNZ_HIP_GRAPH"] = "0"
plans, cfg, dj = nnunet_plans(2, (512, 512), batch_size=2)
tr = Z.nnUNetTrainerSwT2Net(plans, cfg, 0, dj, device=torch.device("cuda"))
tr.use_hip_graph = False
tr.initialize()
b = synthetic_batch(2, (512, 512), tr._get_deep_supervision_scales(), seed=1)
for _ in range(3):
    tr.train_step(b)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.train_step(b)
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.device_time_total)[:14]
for e in rows:
    print(f"{e.device_time_total/1e3:9.2f} ms  n={e.count:4d}  {e.key[:60]:60s} {str(e.input_shapes)[:110]}")
