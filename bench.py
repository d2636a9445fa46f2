#!/usr/bin/env python3
"""bench.py - training patches/s of the nnU-Net hot path on MI355X (BASELINE.json metric, config[1]).

One "step" = nnUNetTrainer.train_step on one batch of synthetic 1x128^3 patches (batch 2 per GPU): forward of the
3d_fullres PlainConvUNet, deep-supervision Dice+CE loss, backward, GradScaler unscale, clip_grad_norm_(12), SGD
step, loss read-back - the step of /root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainer.py:1112-1144.
Inputs are resident in HBM when the timed region starts.

    python bench.py --gpus N --steps K --warmup W
(N > 1: launched by torch.distributed.run, one rank per GPU, RCCL; weak scaling: 2 patches per GPU.)
Prints ONE JSON line on rank 0 (contract in the task description) carrying `roofline` (dominant kernel =
conv_box_kernel, MFMA-bound; algorithmic FLOPs / HIP-event time of its launches inside the timed region) and
`cpu_baseline` (the CPU oracle timed on this host's cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA ~2.5 PF dense"


def cpu_baseline(edge: int = 64):
    """One full training step of the CPU oracle (reference-equivalent torch-CPU path: fp32, no autocast, all host
    threads as run_training.py:256-260 does for -device cpu) on ONE patch of edge^3, scaled to 128^3 by voxels."""
    import multiprocessing
    from oracle.plain_conv_unet import OraclePlainConvUNet, planner_arch_kwargs
    from oracle.losses import deep_supervision_loss
    from nnuzoo_amd.synthetic import synthetic_batch
    from nnuzoo_amd.utilities.network_initialization import InitWeights_He
    cores = multiprocessing.cpu_count()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    net = OraclePlainConvUNet(1, num_classes=2, **planner_arch_kwargs(3, 6, [32, 64, 128, 256, 320, 320]))
    net.apply(InitWeights_He(1e-2))
    opt = torch.optim.SGD(net.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)
    scales = [[1 / 2 ** i] * 3 for i in range(5)]

    def step(e):
        b = synthetic_batch(1, (e, e, e), scales, seed=7)
        opt.zero_grad(set_to_none=True)
        out = net(b['data'])
        l = deep_supervision_loss(out, b['target'], batch_dice=False)
        l.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 12)
        opt.step()
        return float(l)

    # thread-pool / allocator warm-up on a few small convolutions: a full warm-up step would double the ~100 s this
    # sample takes (64^3 is the smallest cube the 6-stage net accepts in training); primitive set-up is < 2 % of the step
    with torch.no_grad():
        w = torch.randn(32, 32, 3, 3, 3)
        for _ in range(3):
            torch.nn.functional.conv3d(torch.randn(1, 32, 32, 32, 32), w, padding=1)
    t0 = time.perf_counter()
    step(edge)
    dt = time.perf_counter() - t0
    patches_per_s = (1.0 / dt) * (edge / 128.0) ** 3
    return {"value": patches_per_s, "unit": "patches/s", "cores": cores, "kind": "port",
            "sample": f"1 full fp32 train step (fwd+loss+bwd+clip+SGD) of the CPU oracle on one {edge}^3 patch "
                      f"({dt:.1f} s), scaled to 128^3 by voxel count; torch {torch.__version__} CPU, {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--patch", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-launch-timer", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with --nproc-per-node {a.gpus} "
                         f"(WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    force_ddp = os.environ.get("NNZ_BENCH_FORCE_DDP") == "1"  # exercise the RCCL reducer path even at world size 1
    if world > 1 or force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from nnuzoo_amd import hip_ops
    from nnuzoo_amd.synthetic import conv_flops_forward, nnunet_plans, synthetic_batch
    from nnuzoo_amd.training.nnUNetTrainer import nnUNetTrainer

    per_gpu_batch = 2
    patch = (a.patch,) * 3
    plans, cfg, dataset_json = nnunet_plans(3, patch, batch_size=per_gpu_batch * world)
    torch.manual_seed(1234)
    trainer = nnUNetTrainer(plans, cfg, 0, dataset_json, device=torch.device("cuda"))
    trainer.initialize()
    assert trainer.batch_size == per_gpu_batch
    batch = synthetic_batch(per_gpu_batch, patch, trainer._get_deep_supervision_scales(), seed=1234 + rank)
    dev = trainer.device
    batch = {"data": batch["data"].to(dev), "target": [t.to(dev) for t in batch["target"]], "keys": batch["keys"]}

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    losses = []
    for _ in range(a.warmup):
        losses.append(float(trainer.train_step(batch)["loss"]))
    hip_ops.TIMER.enabled = not a.no_launch_timer
    hip_ops.TIMER.records = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses.append(float(trainer.train_step(batch)["loss"]))
    barrier()
    dt = time.perf_counter() - t0
    hip_ops.TIMER.enabled = False
    if dist.is_initialized():
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if not all(np.isfinite(losses)):
        raise SystemExit(f"non-finite loss in bench: {losses}")

    if rank == 0:
        patches = per_gpu_batch * world * a.steps
        arch = plans["configurations"][cfg]["architecture"]["arch_kwargs"]
        fwd_flops = sum(conv_flops_forward(arch, patch).values())
        summ = hip_ops.TIMER.summary()
        roof = None
        if "conv_box_kernel" in summ:
            n, fl, sec = summ["conv_box_kernel"]
            ach = fl / sec / 1e12
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "conv_box_kernel_hbm_traffic.json")
            if os.path.exists(tpath):
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            roof = {"bound": "mfma", "kernel": "conv_box_kernel (fprop + dgrad launches of the step)",
                    "achieved": round(ach, 2), "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_F16_DENSE_PEAK_TFLOPS, 4), "traffic": traffic,
                    "launches_per_step": n // a.steps, "avg_launch_us": round(sec / n * 1e6, 2),
                    "flops_per_launch": fl / n}
            if "conv_wgrad_kernel" in summ:
                n2, fl2, sec2 = summ["conv_wgrad_kernel"]
                roof["wgrad_kernel_achieved"] = round(fl2 / sec2 / 1e12, 2)
        line = {
            "metric": "training patches/sec, 3D nnUNet (PlainConvUNet 3d_fullres) 1x128^3 patches",
            "value": round(patches / dt, 3), "unit": "patches/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16 (fp32 accumulate, GradScaler)",
            "data": "synthetic",
            "config": {"workload": f"nnUNet 3d_fullres, synthetic 1x{a.patch}^3 patches, batch {per_gpu_batch}/GPU, "
                                   f"6 stages 32-320 feat, deep supervision, full train_step",
                       "global_batch": per_gpu_batch * world, "parallelism": f"dp{world}",
                       "conv_gflop_per_sample_fwd": round(fwd_flops / 1e9, 1)},
            "patches_per_s_per_gpu": round(patches / dt / world, 3),
            "final_loss": round(losses[-1], 5),
            "roofline": roof,
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
