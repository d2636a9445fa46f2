#!/usr/bin/env python3
"""bench.py - training patches/s of the nnU-Net hot path on MI355X (BASELINE.json metric: "training patches/sec/GPU +
Dice vs ref, 3D nnUNet 128^3 & SS2D2Net 512^2").

Primary leg (the contract's `value`, BASELINE configs[1]): one "step" = nnUNetTrainer.train_step on one batch of
synthetic 1x128^3 patches (batch 2 per GPU): forward of the 3d_fullres PlainConvUNet, deep-supervision Dice+CE loss,
backward, GradScaler unscale, clip_grad_norm_(12), SGD step, loss read-back - the step of
/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainer.py:1112-1144.  Inputs are resident in HBM when the timed
region starts.

Secondary leg (`secondary`, BASELINE configs[2], N = 1 only): nnUNetTrainerM2Net.train_step (M2Net = SS2D^2Net,
/root/reference/nnunetv2/nets/m2net.py:805-971, trainer nnUNetTrainerM2Net.py) on synthetic 1x512^2 patches, batch 2,
forward+loss+backward replayed as one hipGraph (the trainer's default; optimizer tail eager), with the roofline of its
dominant kernel family (the cross-scan backward, HBM-bound by SURVEY.md 8d) timed over eager steps.

    python bench.py --gpus N --steps K --warmup W
N > 1, weak scaling (2 patches per GPU), one process per GPU over RCCL.  Either an outer launcher started the ranks
(`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`: WORLD_SIZE == N in the environment) or -
typed as above - this process starts them itself: BEFORE anything touches the GPU it runs that very launcher as a child
process, relays rank 0's JSON line and exits with the children's status (the reference starts its ranks with mp.spawn,
/root/reference/nnunetv2/run/run_training.py:218-232; a process that has initialised HIP is never re-exec'ed).
NNZ_BENCH_DRYRUN=1 (CPU, gloo, kernel launches stubbed by tests/dryrun.py) exercises launch + process group + the
gradient reducer on the real backward schedule without a GPU; its line says "dryrun": true and carries no throughput.
Prints ONE JSON line on rank 0 carrying `roofline` (dominant kernel conv_box_kernel, MFMA-bound; algorithmic FLOPs /
HIP-event time of its launches inside the timed region), `cpu_baseline` (the CPU oracle, warmed, timed on this host's
cores on a bounded sample), `secondary` and `dice` (protocol results of record).
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

MFMA_F32_DENSE_PEAK_TFLOPS = 157.0   # same guide: fp32 matrix cores (v_mfma_f32_32x32x2_f32)
MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA ~2.5 PF dense"
HBM_PEAK_GBS = 8000.0                # same guide: HBM3E ~8 TB/s


def host_threads():
    """threads for the CPU legs: the physical cores this process may use (SMT siblings only oversubscribe the fp32 conv
    kernels - round 1 timed a cold, 256-thread step and understated the CPU path tenfold)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        smt = open("/sys/devices/system/cpu/smt/active").read().strip() == "1"
    except OSError:
        smt = False
    return max(1, n // 2 if smt else n), n


def cpu_baseline(edge: int = 64, timed_steps: int = 2, budget_s: float = None):
    """Full training steps of the CPU oracle (reference-equivalent torch-CPU path: fp32, no autocast) on ONE
    patch.  First at edge^3 (one untimed warm-up step, `timed_steps` timed ones) - seconds; then at the metric's own 128^3 unless
    even the worst case (8 x the small step) would exceed twice `budget_s` (NNZ_CPU_BASELINE_BUDGET_S, default 120): one step,
    and a second, timed one if it still fits the budget.  The value is then MEASURED at 128^3, not scaled.
    Otherwise the small size scaled by voxel count, as before."""
    from oracle.plain_conv_unet import OraclePlainConvUNet, planner_arch_kwargs
    from oracle.losses import deep_supervision_loss
    from nnuzoo_amd.synthetic import synthetic_batch
    from nnuzoo_amd.utilities.network_initialization import InitWeights_He
    if budget_s is None:
        budget_s = float(os.environ.get("NNZ_CPU_BASELINE_BUDGET_S", "120"))
    threads, logical = host_threads()
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    net = OraclePlainConvUNet(1, num_classes=2, **planner_arch_kwargs(3, 6, [32, 64, 128, 256, 320, 320]))
    net.apply(InitWeights_He(1e-2))
    opt = torch.optim.SGD(net.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)
    scales = [[1 / 2 ** i] * 3 for i in range(5)]

    def run(e, n):
        b = synthetic_batch(1, (e, e, e), scales, seed=7)

        def step():
            opt.zero_grad(set_to_none=True)
            out = net(b['data'])
            l = deep_supervision_loss(out, b['target'], batch_dice=False)
            l.backward()
            torch.nn.utils.clip_grad_norm_(net.parameters(), 12)
            opt.step()
            return float(l)

        t0 = time.perf_counter()
        step()
        warm = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        return warm, (time.perf_counter() - t0) / n

    warm, dt = run(edge, timed_steps)
    tail = f"torch {torch.__version__} CPU, {threads} threads ({logical} logical CPUs visible)"
    small = (f"one {edge}^3 patch: 1 warm-up step ({warm:.1f} s) + {timed_steps} timed steps ({dt:.2f} s each)")
    # The 128^3 step does not cost 8 x the 64^3 one on a many-core host (13.8 s against 6.3 s on 128 threads: the small step is
    # dominated by per-operator overheads), so the guard is the WORST case - voxel scaling - against twice the budget, and the
    # 128^3 warm-up step is itself a measurement: if no second step fits the budget after it, it is the value.
    if edge < 128 and dt * (128.0 / edge) ** 3 <= 2 * budget_s:
        t_begin = time.perf_counter()
        b = synthetic_batch(1, (128, 128, 128), scales, seed=7)

        def step128():
            opt.zero_grad(set_to_none=True)
            l = deep_supervision_loss(net(b['data']), b['target'], batch_dice=False)
            l.backward()
            torch.nn.utils.clip_grad_norm_(net.parameters(), 12)
            opt.step()
            return float(l)

        t0 = time.perf_counter()
        step128()
        warm128 = time.perf_counter() - t0
        if (time.perf_counter() - t_begin) + 1.1 * warm128 <= budget_s:
            t0 = time.perf_counter()
            step128()
            dt128 = time.perf_counter() - t0
            how = f"1 warm-up step ({warm128:.1f} s) + 1 timed step ({dt128:.2f} s)"
        else:
            dt128 = warm128
            how = f"one step ({warm128:.1f} s, the first at this size: a second one did not fit the {budget_s:.0f} s budget)"
        return {"value": 1.0 / dt128, "unit": "patches/s", "cores": threads, "kind": "port",
                "sample": f"CPU oracle, full fp32 train step (fwd+loss+bwd+clip+SGD) on one 128^3 patch - the metric's own size, "
                          f"measured, not scaled: {how}; before it "
                          f"{small}, which scaled by voxels would have given {(1.0 / dt) * (edge / 128.0) ** 3:.4f} patches/s; "
                          + tail}
    return {"value": (1.0 / dt) * (edge / 128.0) ** 3, "unit": "patches/s", "cores": threads, "kind": "port",
            "sample": f"CPU oracle, full fp32 train step (fwd+loss+bwd+clip+SGD) on {small}, scaled to 128^3 by voxel count (even one "
                      f"128^3 step would not fit twice the {budget_s:.0f} s budget on this host); " + tail}


def cpu_scan_baseline(L: int = 8192):
    """SS2D^2Net has no CPU implementation in the reference (its scan is a CUDA extension): the comparator
    is the semantics of selective_scan_ref - a time loop over L - timed forward-only on one call shape and extrapolated
    linearly in state updates to M2Net's 80 calls per forward (5.4 G (b, d, n, t) updates per sample).  An upper bound
    for any CPU step: the rest of the network and the backward are not in it."""
    from oracle.selective_scan import selective_scan_torch
    threads, _ = host_threads()
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    Bt, K, D, N = 2, 4, 32, 16
    u = torch.randn(Bt, K * D, L, generator=g)
    delta = torch.randn(Bt, K * D, L, generator=g) * 0.5
    A = -torch.rand(K * D, N, generator=g) - 0.1
    Bm, Cm = torch.randn(Bt, K, N, L, generator=g), torch.randn(Bt, K, N, L, generator=g)
    with torch.no_grad():
        selective_scan_torch(u[..., :256], delta[..., :256], A, Bm[..., :256], Cm[..., :256], None, None, True)
        t0 = time.perf_counter()
        selective_scan_torch(u, delta, A, Bm, Cm, torch.ones(K * D), torch.zeros(K * D), True)
        dt = time.perf_counter() - t0
    updates = Bt * K * D * N * L
    rate = updates / dt
    per_sample_fwd = 5.4e9 / rate
    return {"value": 1.0 / per_sample_fwd, "unit": "patches/s", "cores": threads, "kind": "port",
            "sample": f"oracle selective scan (selective_scan_ref semantics), forward only, (2, 128, 16, {L}) in {dt:.2f} s = "
                      f"{rate / 1e6:.1f} M state-updates/s; extrapolated to the 5.4 G updates of one M2Net 512^2 forward "
                      f"sample - scan forward ONLY, an upper bound on any CPU step rate"}


def cpu_m2net_step_baseline(budget_s: float = None):
    """Full fp32 training step of the CPU oracle of M2Net (oracle/m2net.py, pinned by the reference's own whole-net outputs and
    gradients: tests/test_oracle_m2net.py) - forward, deep-supervision Dice + CE, backward, clip, AdamW - on ONE patch: first at
    64^2, then ("full step at 128^2") at 128^2 when 7 x the 64^2 time fits `budget_s` (NNZ_CPU_BASELINE_BUDGET_S,
    default 120).  `value` extrapolates the largest size run to the 512^2 patch of the metric linearly in pixels - the scan's
    time loop, which dominates, is linear in L; with that caveat."""
    from oracle import m2net as om
    from oracle.losses import deep_supervision_loss
    from nnuzoo_amd.synthetic import synthetic_batch
    if budget_s is None:
        budget_s = float(os.environ.get("NNZ_CPU_BASELINE_BUDGET_S", "120"))
    threads, logical = host_threads()
    threads = min(threads, 16)     # thousands of small tensor operations per scan: more threads only add fork / join time
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    net = om.M2Net(1, 2, True).train()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=5e-2, eps=1e-5)
    scales = [[1, 1]] + [[1 / 2 ** i] * 2 for i in range(6)]

    def step(size):
        b = synthetic_batch(1, (size, size), scales, seed=7)
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss = deep_supervision_loss(list(net(b["data"])), b["target"], batch_dice=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 12)
        opt.step()
        return time.perf_counter() - t0

    times = {64: step(64)}
    if 7.0 * times[64] <= budget_s:
        times[128] = step(128)
    size = max(times)
    dt = times[size]
    return {"value": (1.0 / dt) * (size / 512.0) ** 2, "unit": "patches/s", "cores": threads, "kind": "port",
            "sample": "CPU oracle of M2Net (plain torch fp32, selective_scan_ref semantics as a time loop), full train step "
                      "(fwd + deep-supervision DC+CE + bwd + clip + AdamW) on one patch: "
                      + ", ".join(f"{k}^2 in {v:.1f} s" for k, v in sorted(times.items()))
                      + f"; value = the {size}^2 step extrapolated to one 512^2 patch linearly in pixels (the scan's time loop "
                      f"dominates and is linear in L); torch {torch.__version__} CPU, {threads} threads ({logical} logical CPUs)",
            "step_seconds": {str(k): round(v, 2) for k, v in times.items()}}


def cpu_swt2net_step_baseline(timed_steps: int = 2):
    """Full fp32 training step of the CPU oracle of SwT2Net (oracle/swt2net.py, pinned by the reference's own whole-net outputs
    and autograd: tests/test_oracle_swt2net.py) on the `swt2net` leg's OWN workload - batch 2 of 512^2, forward, deep-supervision
    DC + CE, backward, clip, AdamW - one warm-up step, then `timed_steps` timed ones.  No extrapolation."""
    from oracle.swt2net import SwT2Net
    from oracle.losses import deep_supervision_loss
    from nnuzoo_amd.synthetic import synthetic_batch
    threads, logical = host_threads()
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    net = SwT2Net(1, 2, True).train()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=5e-2, eps=1e-5)
    b = synthetic_batch(2, (512, 512), [[1, 1]] + [[1 / 2 ** i] * 2 for i in range(6)], seed=7)

    def step():
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss = deep_supervision_loss(list(net(b["data"])), b["target"], batch_dice=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 12)
        opt.step()
        return time.perf_counter() - t0

    warm = step()
    dt = sum(step() for _ in range(timed_steps)) / timed_steps
    return {"value": 2.0 / dt, "unit": "patches/s", "cores": threads, "kind": "port",
            "sample": f"CPU oracle of SwT2Net (plain torch fp32), full train step (fwd + deep-supervision DC+CE + bwd + clip + "
                      f"AdamW) on the leg's own batch of 2 x 512^2: 1 warm-up step ({warm:.1f} s) + {timed_steps} timed steps "
                      f"({dt:.2f} s each); torch {torch.__version__} CPU, {threads} threads ({logical} logical CPUs)"}


def _profile_json(name):
    path = os.path.join(ROOT, "profiles", name)
    return json.load(open(path)) if os.path.exists(path) else None


def run_secondary(steps: int, warmup: int):
    """BASELINE configs[2]: M2Net (SS2D^2Net) 1x512^2, batch 2, nnUNetTrainerM2Net.train_step (autocast step, AdamW)"""
    from nnuzoo_amd import backends as _bk
    from nnuzoo_amd import hip_ops
    from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
    from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerM2Net
    size, batch = 512, 2
    plans, cfg, dj = nnunet_plans(2, (size, size), batch_size=batch)
    torch.manual_seed(0)
    tr = nnUNetTrainerM2Net(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    b = synthetic_batch(batch, (size, size), tr._get_deep_supervision_scales(), seed=3)
    b = {"data": b["data"].cuda(), "target": [t.cuda() for t in b["target"]]}
    losses = []
    graph = bool(getattr(tr, "use_hip_graph", False))
    for _ in range(warmup):
        losses.append(float(tr.train_step(b)["loss"]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses.append(float(tr.train_step(b)["loss"]))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # roofline leg: per-launch HIP-event timing needs the launches to be issued from Python, which a replayed hipGraph
    # does not do - the same kernels are timed over a few EAGER steps right after the timed region
    roof_steps = min(5, steps)
    tr.use_hip_graph = False
    tr.train_step(b)
    hip_ops.TIMER.enabled = True
    hip_ops.TIMER.records = []
    for _ in range(roof_steps):
        losses.append(float(tr.train_step(b)["loss"]))
    torch.cuda.synchronize()
    hip_ops.TIMER.enabled = False
    summ = hip_ops.TIMER.summary()
    hip_ops.TIMER.records = []
    if not all(np.isfinite(losses)):
        raise SystemExit(f"non-finite loss in the secondary bench: {losses}")
    roof = None
    if "ss2d_scan_bwd" in summ:
        n, by, sec = summ["ss2d_scan_bwd"]
        ach = by / sec / 1e9
        tjs = _profile_json("ss2d_scan_bwd_hbm_traffic.json")
        roof = {"bound": "hbm", "kernel": "ss2d cross-scan backward (summary + carry + final + finalize kernels per call: xs_rl_bwd_* for >= 2 M row-steps, scan_bwd_kernel below)",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": tjs.get("hbm_bytes_per_launch") if tjs else None,
                "traffic_source": ("profiles/ss2d_scan_bwd_hbm_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                   "over one eager M2Net step, all kernels of the cross-scan backward per call, "
                                   "tools/pmc_traffic.py; re-measured by tools/collect_profiles.sh every round)") if tjs else None,
                "algorithmic_bytes_per_launch": round(by / n),
                "launches_per_step": n // roof_steps, "avg_launch_us": round(sec / n * 1e6, 2),
                "bytes_per_launch": by / n, "ms_per_step": round(sec / roof_steps * 1e3, 3),
                "timed_over": f"{roof_steps} eager steps after the timed region (same kernels; a replayed graph issues no "
                              f"per-launch events)"}
        if "ss2d_scan_fwd" in summ:
            n2, by2, sec2 = summ["ss2d_scan_fwd"]
            roof["fwd_achieved"] = round(by2 / sec2 / 1e9, 1)
            roof["fwd_ms_per_step"] = round(sec2 / roof_steps * 1e3, 3)
    out = {"metric": "training patches/sec, SS2D2Net (M2Net) 1x512^2 patches", "value": round(batch * steps / dt, 3),
           "unit": "patches/s", "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3),
           "dtype": "f32 scan / f16 autocast elsewhere (GradScaler)",
           "config": {"workload": "M2Net (SS2D^2Net) 2d, synthetic 1x512^2 patches, batch 2, deep supervision, full "
                                  "nnUNetTrainerM2Net.train_step (fused AdamW); forward+loss+backward "
                                  + ("replayed as one hipGraph" if graph else "eager")},
           "hip_graph": graph,
           "final_loss": round(losses[-1], 5), "roofline": roof, "backends": _bk.report(tr.network),
           "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}
    dz = None
    if os.environ.get("NNZ_BENCH_LIVE_DICE", "1") != "0":
        # the Dice protocol of this path, run HERE (not a number read from a file): the HIP M2NetP trains the
        # 60 fp32 steps of the committed CPU-oracle run (tests/golden/dice_oracle_m2netp_64.json: inputs, the oracle's losses,
        # Dice and masks - data, written by tools/dice_oracle_cpu_zoo.py in the build container) and is scored on the same
        # held-out patches; ~10 s
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from dice_parity_zoo import run_vs_oracle
            t0 = time.perf_counter()
            dz = run_vs_oracle(os.path.join(ROOT, "tests", "golden", "dice_oracle_m2netp_64.json"))
            out["dice"] = {"hip": round(dz["dice_hip"], 5), "cpu_oracle": round(dz["dice_oracle"], 5),
                           "abs_delta": round(dz["abs_delta"], 5), "mask_agreement": round(dz["mask_agreement"], 5),
                           "loss_abs_delta_step0": float(f"{dz['loss_abs_delta_step0']:.3g}"),
                           "measured_in_this_run": True, "seconds": round(time.perf_counter() - t0, 1),
                           "source": "tools/dice_parity_zoo.run_vs_oracle inside this bench run: HIP M2NetP vs the CPU oracle "
                                     "oracle/m2net.py (fixture tests/golden/dice_oracle_m2netp_64.json), 64^2, 60 identical fp32 "
                                     "steps, Dice on 16 held-out patches; gate of tests/test_dice_parity_zoo_gpu.py: 0.01"}
        except Exception as e:      # the protocol must not take the throughput line down with it
            out["dice_error"] = repr(e)[:200]
            dz = None
    if dz is None:
        dz = _profile_json("r06_dice_m2netp_64_vs_oracle.json") or _profile_json("r05_dice_m2netp_64_vs_oracle.json")
    if dz and "dice" not in out:
        out["dice"] = {"hip": round(dz["dice_hip"], 5), "cpu_oracle": round(dz["dice_oracle"], 5),
                       "abs_delta": round(dz["abs_delta"], 5), "mask_agreement": round(dz["mask_agreement"], 5),
                       "measured_in_round": 6 if _profile_json("r06_dice_m2netp_64_vs_oracle.json") else 5,
                       "source": "profiles/r0N_dice_m2netp_64_vs_oracle.json (tools/dice_parity_zoo.py --oracle-json "
                                 "tests/golden/dice_oracle_m2netp_64.json: HIP M2NetP vs the CPU oracle oracle/m2net.py, 64^2, 60 "
                                 "identical fp32 steps; protocol result of record, re-run by tests/test_dice_parity_zoo_gpu.py, "
                                 "not inside this bench run)"}
    del tr
    torch.cuda.empty_cache()
    return out


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` typed without a launcher: start the N ranks as CHILD processes through
    torch.distributed.run (rendezvous on 127.0.0.1, a free port), pass their output through and return their status."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for ln in p.stdout:
        sys.stdout.write(ln)
        sys.stdout.flush()
    return p.wait()


def dryrun(a, world: int, rank: int) -> None:
    """Plumbing check without a GPU (tests/test_bench_launch.py): the real forward / backward SCHEDULE of the 3d_fullres
    network with every kernel launch replaced by a host stand-in (tests/dryrun.py), under a gloo process group - the same
    launcher, environment handling, trainer set-up (batch split, parameter broadcast, reducer attachment), bucketed
    in-place all-reduce of the gradient arena and JSON line as the GPU run.  Measures nothing."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dryrun import stub_kernel_launches
    from nnuzoo_amd.synthetic import nnunet_plans
    from nnuzoo_amd.training.nnUNetTrainer import nnUNetTrainer
    if os.environ.get("NNZ_BENCH_DRYRUN_FAIL") == "1" and rank == world - 1:
        raise SystemExit(3)                    # test hook: a failing rank must become the parent's exit status
    if world > 1:
        dist.init_process_group("gloo")
    plans, cfg, dataset_json = nnunet_plans(3, (128,) * 3, batch_size=2 * world)
    torch.manual_seed(1234 + rank)             # different per rank: attach_bucketed_allreduce must broadcast rank 0's
    tr = nnUNetTrainer(plans, cfg, 0, dataset_json, device=torch.device("cpu"))
    tr.initialize()
    assert tr.batch_size == 2
    net = tr.network
    x = torch.zeros(tr.batch_size, 1, 32, 32, 32)
    for _ in range(a.warmup + a.steps):
        with stub_kernel_launches(float(rank + 1)):
            outs, rec = net._run_forward(x, save=True)
            gouts = [torch.zeros_like(o) for o in outs]
            gouts[-1] = None                   # the deep-supervision output of weight 0
            net._run_backward(rec, gouts)
    arena, red = net.grad_arena(), net.grad_reducer
    mean = sum(r + 1 for r in range(world)) / world
    ok = bool(((arena == mean) | (arena == 0)).all())   # every reduced gradient is the mean over ranks
    if world > 1:
        t = torch.tensor([float(ok)])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = bool(t.item())
    if rank == 0:
        print(json.dumps({"metric": "training patches/sec, 3D nnUNet (PlainConvUNet 3d_fullres) 1x128^3 patches",
                          "dryrun": True, "value": None, "unit": "patches/s", "n_gpus": world, "steps": a.steps,
                          "warmup": a.warmup, "scaling": "weak", "rccl_ranks": dist.get_world_size() if world > 1 else 0,
                          "backend": "gloo", "allreduce_buckets_per_step": getattr(red, "buckets_last_step", None),
                          "arena_floats": int(arena.numel()), "gradients_are_rank_mean": ok,
                          "config": {"global_batch": 2 * world, "parallelism": f"dp{world}"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_swt2net(steps: int, warmup: int):
    """BASELINE configs[3]: SwT2Net 1x512^2, batch 2, nnUNetTrainerSwT2Net.train_step (fp32 step: no autocast, no GradScaler,
    /root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainerSwT2Net.py:112-130).  roofline = the window-attention core
    (the op BASELINE.json's north_star names as the dense contraction) against the fp32 MFMA peak."""
    from nnuzoo_amd import hip_ops
    from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
    from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerSwT2Net
    size, batch = 512, 2
    plans, cfg, dj = nnunet_plans(2, (size, size), batch_size=batch)
    torch.manual_seed(0)
    tr = nnUNetTrainerSwT2Net(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    b = synthetic_batch(batch, (size, size), tr._get_deep_supervision_scales(), seed=3)
    b = {"data": b["data"].cuda(), "target": [t.cuda() for t in b["target"]]}
    graph = bool(getattr(tr, "use_hip_graph", False))
    losses = [float(tr.train_step(b)["loss"]) for _ in range(warmup)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses.append(float(tr.train_step(b)["loss"]))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    roof_steps = min(3, steps)
    tr.use_hip_graph = False
    tr.train_step(b)
    hip_ops.TIMER.enabled = True
    hip_ops.TIMER.records = []
    for _ in range(roof_steps):
        losses.append(float(tr.train_step(b)["loss"]))
    torch.cuda.synchronize()
    hip_ops.TIMER.enabled = False
    summ = hip_ops.TIMER.summary()
    hip_ops.TIMER.records = []
    if not all(np.isfinite(losses)):
        raise SystemExit(f"non-finite loss in the SwT2Net bench: {losses}")
    roof = None
    if "win_attn_fwd" in summ and "win_attn_bwd" in summ:
        nf, ff, sf = summ["win_attn_fwd"]
        nb, fb, sb = summ["win_attn_bwd"]
        ach = (ff + fb) / (sf + sb) / 1e12
        roof = {"bound": "mfma", "kernel": "win_attn_fwd_kernel + win_attn_bwd_kernel (fp32 v_mfma_f32_32x32x2_f32)",
                "achieved": round(ach, 2), "peak": MFMA_F32_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / MFMA_F32_DENSE_PEAK_TFLOPS, 4), "traffic": None,
                "launches_per_step": (nf + nb) // roof_steps, "fwd_avg_launch_us": round(sf / nf * 1e6, 2),
                "bwd_avg_launch_us": round(sb / nb * 1e6, 2), "ms_per_step": round((sf + sb) / roof_steps * 1e3, 3),
                "fwd_achieved": round(ff / sf / 1e12, 2), "bwd_achieved": round(fb / sb / 1e12, 2),
                "timed_over": f"{roof_steps} eager steps after the timed region"}
        tj = _profile_json("win_attn_hbm_traffic.json")
        if tj:      # HBM bytes per window-attention launch (forward and backward launches together) from the PMC passes of record
            roof["traffic"] = tj.get("hbm_bytes_per_launch")
            roof["traffic_source"] = "profiles/win_attn_hbm_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE " \
                                     "passes over one eager SwT2Net step, tools/pmc_traffic.py; re-measured by tools/collect_profiles.sh every round)"
    from nnuzoo_amd import backends as _bk
    backends = _bk.report(tr.network)      # which kernel family every dispatching module took (nnuzoo_amd/backends.py)
    out = {"metric": "training patches/sec, SwT2Net 1x512^2 patches", "value": round(batch * steps / dt, 3),
           "unit": "patches/s", "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3),
           "dtype": "f32 (no autocast, like the reference trainer)",
           "config": {"workload": "SwT2Net 2d, synthetic 1x512^2 patches, batch 2, deep supervision, full "
                                  "nnUNetTrainerSwT2Net.train_step (fused AdamW); forward+loss+backward "
                                  + ("replayed as one hipGraph" if graph else "eager")},
           "hip_graph": graph, "final_loss": round(losses[-1], 5), "roofline": roof,
           "backends": backends, "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}
    del tr
    torch.cuda.empty_cache()
    return out


_REAL_STDOUT_FD = None


def _emit(text: str):
    """the contract line, on the process's real stdout (see main)"""
    sys.stdout.flush()
    if _REAL_STDOUT_FD is None:
        print(text, flush=True)
        return
    import ctypes
    ctypes.CDLL(None).fflush(None)            # whatever C code buffered for stdout so far leaves through stderr
    os.dup2(_REAL_STDOUT_FD, 1)
    os.write(1, (text + "\n").encode())
    os.dup2(2, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)     # >= 100 timed, >= 20 warm-up iterations
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--patch", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-launch-timer", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--secondary-steps", type=int, default=20)
    ap.add_argument("--secondary-warmup", type=int, default=5)
    ap.add_argument("--no-swt2net", action="store_true")
    ap.add_argument("--no-h2d-leg", action="store_true")
    ap.add_argument("--tune", default="", help="A/B knobs, e.g. norm1=1024,conv3=0 (nnz_norm_tuning / nnz_conv_tuning)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a.gpus))          # nothing has touched the GPU yet
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but the launcher set WORLD_SIZE={world}")
    backend = os.environ.get("NNZ_BENCH_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm
    if os.environ.get("NNZ_BENCH_DRYRUN") == "1":
        return dryrun(a, world, rank)
    # stdout carries ONE JSON line.  Libraries print there too (this RCCL build writes a version banner to C stdout, which
    # surfaced BEHIND the JSON line when the buffer was flushed at exit): everything written to fd 1 during the run goes to
    # stderr instead, the line is printed on the saved descriptor (_emit) and fd 1 points at stderr again afterwards.
    global _REAL_STDOUT_FD
    sys.stdout.flush()
    _REAL_STDOUT_FD = os.dup(1)
    os.dup2(2, 1)
    # NNZ_BENCH_SHARE_GPU=1: several ranks on one device (gloo only - RCCL refuses duplicate devices); used by the
    # 1-GPU box test that runs the real two-rank step
    share = os.environ.get("NNZ_BENCH_SHARE_GPU") == "1"
    local_rank = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(local_rank)
    force_ddp = os.environ.get("NNZ_BENCH_FORCE_DDP") == "1"  # exercise the RCCL reducer path even at world size 1
    if world > 1 or force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from nnuzoo_amd import hip_ops
    from nnuzoo_amd._lib import call as _call
    for kv in [t for t in a.tune.split(",") if t]:
        name, val = kv.split("=")
        fam, knob = ("nnz_norm_tuning", name[4:]) if name.startswith("norm") else ("nnz_conv_tuning", name[4:])
        _call(fam, int(knob), int(val))
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
    # N = 1: forward + loss + backward replayed as one hipGraph; N > 1 (or NNZ_BENCH_FORCE_DDP=1): eager step with the bucketed
    # all-reduce overlapped with the backward (NNZ_DDP_GRAPH=1 opts into graph SEGMENTS with the RCCL collectives between them,
    # training/graph_step.py GraphedDDPStep: +0.4 % at world size 1, see nnUNetTrainer.train_step)
    graph = bool(getattr(trainer, "use_hip_graph", False)) and \
        ((world == 1 and not force_ddp) or os.environ.get("NNZ_DDP_GRAPH", "0") == "1")
    trainer.use_hip_graph = graph
    for _ in range(a.warmup):
        losses.append(float(trainer.train_step(batch)["loss"]))
    # Per-launch HIP events (the roofline leg) are NOT taken inside the timed region: a replayed graph (N = 1) issues none, and on
    # eager steps (N > 1) an event pair around each of ~90 launches costs the step 5 % (147 vs 139 patches/s, same box).  The same
    # kernels are timed over eager steps right after the region, as the SS2D^2Net and SwT2Net legs do.
    hip_ops.TIMER.records = []
    timed_steps_with_events = 0
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        losses.append(float(trainer.train_step(batch)["loss"]))
    barrier()
    dt = time.perf_counter() - t0
    hip_ops.TIMER.enabled = False
    # SURVEY.md 8d counts the H2D of the batch into the step; the contract's `value` is the HBM-resident rate (inputs in HBM
    # when the timed region starts).  Both are measured: a second, shorter region feeds every step from PINNED host buffers
    # (16.8 MB fp32 image + int16 targets per batch; train_step's `.to(device, non_blocking=True)` is the reference's own
    # line, nnUNetTrainer.py:1116-1121) - reported as `h2d_inclusive`, never as `value`.
    h2d = None
    if not a.no_h2d_leg:
        host = {"data": batch["data"].cpu().pin_memory(), "target": [t.cpu().pin_memory() for t in batch["target"]]}
        nh = max(4, a.steps // 4)
        trainer.train_step(host)
        barrier()
        th = time.perf_counter()
        for _ in range(nh):
            losses.append(float(trainer.train_step(host)["loss"]))
        barrier()
        dth = time.perf_counter() - th
        h2d = {"value": round(per_gpu_batch * world * nh / dth, 3), "unit": "patches/s", "steps": nh,
               "ms_per_step": round(dth / nh * 1e3, 3),
               "batch_bytes": int(sum(t.numel() * t.element_size() for t in [host["data"]] + host["target"])),
               "note": "same step fed from pinned host memory every step (H2D inside the timed region, SURVEY.md 8d); max over "
                       "ranks not taken: rank 0's own clock"}
    roof_note = ""
    eager_ms = None
    if not a.no_launch_timer:
        # the per-launch events are taken over eager steps AFTER the timed region at every N (round 6; at N > 1 they used to ride on
        # every 4th step inside it: ~1.2 % of the rate the driver computes its scaling efficiency from).  All ranks run these
        # steps - they contain the same collectives as any other step.
        trainer.use_hip_graph = False
        trainer.train_step(batch)
        timed_steps_with_events = min(8, a.steps)
        hip_ops.TIMER.enabled = True
        torch.cuda.synchronize()
        te = time.perf_counter()
        for _ in range(timed_steps_with_events):
            losses.append(float(trainer.train_step(batch)["loss"]))
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - te) / timed_steps_with_events * 1e3   # incl. the event pairs around every launch
        hip_ops.TIMER.enabled = False
        trainer.use_hip_graph = graph
        roof_note = (f"HIP events around every launch over {timed_steps_with_events} EAGER steps right after the timed region "
                     + ("(the timed region replays the step as one hipGraph: no per-launch events exist there; same kernels, "
                        "same data)" if graph else "(the timed region itself carries no events)"))
    rccl_ranks = dist.get_world_size() if dist.is_initialized() else 0
    reducer = getattr(trainer.network, "grad_reducer", None)
    buckets_per_step = getattr(reducer, "buckets_last_step", None)
    rank_seconds = [dt]
    if dist.is_initialized():
        # max over ranks is the contract's clock; every rank's own time is reported beside it so that a first multi-GPU run shows
        # at a glance whether one rank (its GPU, its xGMI links) lags or all of them pay the all-reduce
        tt = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(tt, torch.tensor([dt], device=dev, dtype=torch.float64))
        rank_seconds = [float(x.item()) for x in tt]
        dt = max(rank_seconds)
    if not all(np.isfinite(losses)):
        raise SystemExit(f"non-finite loss in bench: {losses}")

    if rank == 0:
        patches = per_gpu_batch * world * a.steps
        arch = plans["configurations"][cfg]["architecture"]["arch_kwargs"]
        fwd_flops = sum(conv_flops_forward(arch, patch).values())
        summ = hip_ops.TIMER.summary()
        hip_ops.TIMER.records = []
        roof = None
        if "conv_box_kernel" in summ:
            n, fl, sec = summ["conv_box_kernel"]
            ach = fl / sec / 1e12
            traffic, tsrc = None, None
            tj = _profile_json("conv_box_kernel_hbm_traffic.json")
            if tj:
                traffic, tsrc = tj.get("hbm_bytes_per_launch"), "profiles/conv_box_kernel_hbm_traffic.json (separate " \
                    "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command, tools/pmc_traffic.py)"
            roof = {"bound": "mfma", "kernel": "conv_box_kernel (fprop + dgrad launches of the step)",
                    "achieved": round(ach, 2), "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_F16_DENSE_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": tsrc,
                    "launches_per_step": n // max(1, timed_steps_with_events), "avg_launch_us": round(sec / n * 1e6, 2),
                    "flops_per_launch": fl / n, "ms_per_step": round(sec / max(1, timed_steps_with_events) * 1e3, 3),
                    "timed_over": roof_note}
            if eager_ms is not None:
                # the same step launched eagerly with an event pair around every launch (what the roofline was timed in)
                roof["eager_instrumented_ms_per_step"] = round(eager_ms, 3)
            if "conv_wgrad_kernel" in summ:
                n2, fl2, sec2 = summ["conv_wgrad_kernel"]
                roof["wgrad_kernel_achieved"] = round(fl2 / sec2 / 1e12, 2)
                roof["wgrad_ms_per_step"] = round(sec2 / max(1, timed_steps_with_events) * 1e3, 3)
        line = {
            "metric": "training patches/sec, 3D nnUNet (PlainConvUNet 3d_fullres) 1x128^3 patches",
            "value": round(patches / dt, 3), "unit": "patches/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16 (fp32 accumulate, GradScaler)",
            "data": "synthetic",
            "config": {"workload": f"nnUNet 3d_fullres, synthetic 1x{a.patch}^3 patches HBM-resident when the timed region starts "
                                   f"(the H2D-inclusive rate is `h2d_inclusive`), batch {per_gpu_batch}/GPU, "
                                   f"6 stages 32-320 feat, deep supervision, full train_step"
                                   + ((" (forward+loss+backward replayed as hipGraph segments, RCCL all-reduce between them)"
                                       if getattr(trainer, "_graphed_ddp", None) is not None else
                                       " (forward+loss+backward replayed as one hipGraph)") if graph else " (eager)"),
                       "global_batch": per_gpu_batch * world, "parallelism": f"dp{world}",
                       "conv_gflop_per_sample_fwd": round(fwd_flops / 1e9, 1)},
            "patches_per_s_per_gpu": round(patches / dt / world, 3),
            "final_loss": round(losses[-1], 5),
            "hip_graph": graph, "hip_graph_segments": (len(trainer._graphed_ddp.segments)
                                                        if getattr(trainer, "_graphed_ddp", None) is not None else (1 if graph else 0)),
            "rccl_ranks": rccl_ranks, "allreduce_buckets_per_step": buckets_per_step,
            "allreduce_bytes_per_step": getattr(reducer, "bytes_last_step", None),
            "rank_ms_per_step": [round(x / a.steps * 1e3, 3) for x in rank_seconds],
            "roofline": roof,
            "h2d_inclusive": h2d,
        }
        dz, dround = None, 0
        fx3 = os.path.join(ROOT, "tests", "golden", "dice_oracle_plainconv_64.json")
        if rank == 0 and world == 1 and os.environ.get("NNZ_BENCH_LIVE_DICE", "1") != "0" and os.path.exists(fx3):
            # the Dice protocol of the primary path, run HERE: the HIP PlainConvUNet trains the 100 steps at 64^3
            # of the committed CPU-oracle run (tools/dice_oracle_cpu.py; fixture = the oracle's losses, Dice, masks) and is scored on
            # the same held-out patches; ~15 s
            try:
                trainer = None                 # (its graph and pools go before a second trainer is built)
                torch.cuda.empty_cache()
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                from dice_parity import run_vs_oracle as _dice3d
                t0d = time.perf_counter()
                d3 = _dice3d(fx3)
                line["dice"] = {"hip": round(d3["dice_hip"], 5), "oracle": round(d3["dice_oracle"], 5),
                                "abs_delta": round(d3["abs_delta"], 6), "mask_agreement": round(d3["mask_agreement"], 5),
                                "loss_abs_delta_step0": float(f"{d3['loss_abs_delta_step0']:.3g}"),
                                "measured_in_this_run": True, "seconds": round(time.perf_counter() - t0d, 1),
                                "source": "tools/dice_parity.run_vs_oracle inside this bench run: HIP PlainConvUNet (product train_step) "
                                          "vs the CPU oracle oracle/plain_conv_unet.py (fixture tests/golden/dice_oracle_plainconv_64.json), "
                                          "64^3, 100 identical steps, Dice on 16 held-out patches; gate of tests/test_dice_parity_gpu.py: 0.01"}
            except Exception as e:
                line["dice_error"] = repr(e)[:200]
        for dround in (6, 5, 4, 3, 2, 1):            # otherwise the newest protocol result on file; the round it was measured in is reported
            if "dice" in line:
                break
            dz = _profile_json(f"r0{dround}_dice_parity_64cubed.json")
            if dz:
                break
        if dz and "dice" not in line:
            line["dice"] = {"hip": round(dz["dice_hip"], 5), "oracle": round(dz["dice_oracle"], 5),
                            "abs_delta": round(dz["abs_delta"], 6), "mask_agreement": round(dz["mask_agreement"], 5),
                            "measured_in_round": dround,
                            "source": "profiles/*dice_parity_64cubed.json (tools/dice_parity.py: 100 identical steps at "
                                      "64^3 on HIP and on the CPU oracle, 16 held-out patches; protocol result of "
                                      "record, not re-run here - the CPU side takes ~10 min)"}
    del trainer, batch
    torch.cuda.empty_cache()
    if rank == 0:
        if world == 1 and not a.no_secondary:
            line["secondary"] = run_secondary(a.secondary_steps, a.secondary_warmup)
            if not a.no_swt2net:
                line["swt2net"] = run_swt2net(max(4, a.secondary_steps // 2), max(2, a.secondary_warmup // 2))
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            if "secondary" in line:
                # the full oracle step (128^2 when the host finishes it inside the budget) with the forward-only scan figure
                # at the metric's own sequence lengths beside it (SURVEY.md 8d asks for both)
                sec = cpu_m2net_step_baseline()
                sec["scan_forward_only"] = cpu_scan_baseline()
                line["secondary"]["cpu_baseline"] = sec
            if "swt2net" in line:
                line["swt2net"]["cpu_baseline"] = cpu_swt2net_step_baseline()
        # the zoo legs' headline numbers as TOP-LEVEL scalars (a parser that keeps only scalars of the contract
        # line still records the SS2D^2Net half of BASELINE's metric and the SwT2Net leg)
        for key in ("secondary", "swt2net"):
            leg = line.get(key)
            if isinstance(leg, dict):
                line[f"{key}_value"] = leg.get("value")
                line[f"{key}_unit"] = leg.get("unit")
                line[f"{key}_ms_per_step"] = leg.get("ms_per_step")
                if isinstance(leg.get("roofline"), dict):
                    line[f"{key}_roofline_frac"] = leg["roofline"].get("frac")
                if isinstance(leg.get("dice"), dict):
                    line[f"{key}_dice_abs_delta"] = leg["dice"].get("abs_delta")
        if isinstance(line.get("roofline"), dict):
            line["roofline_frac"] = line["roofline"].get("frac")
        if isinstance(line.get("dice"), dict):
            line["dice_abs_delta"] = line["dice"].get("abs_delta")
        _emit(json.dumps(line))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
