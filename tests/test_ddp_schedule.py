"""Data-parallel readiness of the REAL backward schedule without GPUs (CPU, gloo, world size 2).

`PlainConvUNet._run_backward` (nnuzoo_amd/nets/plain_conv_unet.py) allocates every parameter gradient from one flat
arena in completion order and calls `grad_reducer.stage_done_arena` after each decoder / encoder stage; the reducer
all-reduces contiguous arena slices in place while the next stage computes.  Here the schedule itself runs - the real
`_run_forward` / `_run_backward` of the 6-stage 3d_fullres network - with only the kernel LAUNCHES replaced by host
stand-ins that fill the gradient buffers with a rank-dependent constant (tests/dryrun.py).  Checked:
  * the slice boundaries are the same on every rank and every step, contiguous, and cover the arena exactly once;
  * with the default 12 MB buckets there are >= 6 collectives, the last one (what cannot overlap with backward) is small
    enough that its ring time over xGMI (2 * 7/8 * bytes / 153 GB/s) is below the time of the last weight-gradient kernel
    of the step (enc0.1 wgrad 0.28 ms, profiles/r01_conv_layers_v6.txt);
  * after `finish_arena` every used gradient equals the mean over ranks, unused heads stay zero / None;
  * arena order = reverse completion order and is identical across ranks (layout is a pure function of the schedule).
Reference behaviour being replaced: torch DDP wrapping at nnUNetTrainer.py:278-280 (bucketed NCCL all-reduce overlapped
with backward by autograd hooks)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dryrun import stub_kernel_launches  # tests/dryrun.py


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build_net():
    from nnuzoo_amd.nets.plain_conv_unet import PlainConvUNet
    from nnuzoo_amd.synthetic import nnunet_plans
    from nnunetv2.utilities.get_network_from_plans import get_network_from_plans
    plans, cfg, _ = nnunet_plans(3, (128, 128, 128))
    arch = plans["configurations"][cfg]["architecture"]
    net = get_network_from_plans(arch["network_class_name"], arch["arch_kwargs"], arch["_kw_requires_import"], 1, 2,
                                 allow_init=True, deep_supervision=True)
    assert isinstance(net, PlainConvUNet)
    return net


def _one_step(net, fill):
    """the real forward + backward schedule at 1 x 32^3 (arena layout does not depend on the patch size)"""
    x = torch.zeros(1, 1, 32, 32, 32)
    with stub_kernel_launches(fill):
        outs, rec = net._run_forward(x, save=True)
        gouts = [torch.zeros_like(o) for o in outs]
        gouts[-1] = None                               # deep-supervision weight 0: no gradient for the last output
        grads = net._run_backward(rec, gouts)
    return grads


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nnuzoo_amd.ddp import attach_bucketed_allreduce
        torch.manual_seed(10 + rank)
        net = _build_net()
        red = attach_bucketed_allreduce(net)            # default bucket size: what the trainer uses
        first = [p.detach().flatten()[:4].tolist() for p in net.parameters()]
        gathered = [None] * world
        dist.all_gather_object(gathered, first)
        assert gathered[0] == gathered[1]               # rank 0's parameters everywhere
        record = []
        for step in range(2):
            grads = _one_step(net, fill=float(rank + 1))
            arena = net.grad_arena()
            total = sum(p.numel() for p in net.parameters())
            assert arena.numel() == total
            sl = list(red.slices_last_step)
            assert sl[0][0] == 0 and sl[-1][1] == total and all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
            record.append(sl)
            unused = net.grad_arena_unused()
            mean = sum(r + 1 for r in range(world)) / world
            layout = net.grad_arena_layout()
            assert [id(p) for p, _ in layout] != [id(p) for p in net.parameters()]      # completion order, not registration
            for (p, off), g in zip(layout, [None] * len(layout)):
                seg = arena[off:off + p.numel()]
                if id(p) in unused:
                    assert float(seg.abs().max()) == 0.0
                elif p.dim() == 1 and any(p is m.conv.bias for m in _conv_blocks(net)):
                    assert float(seg.abs().max()) == 0.0      # bias in front of InstanceNorm: identically zero gradient
                else:
                    assert torch.allclose(seg, torch.full_like(seg, mean)), (off, seg[:3], mean)
            by_param = dict(zip(net.parameters(), grads))
            assert sum(g is None for g in grads) == 2 and all(by_param[p] is None for p in net.parameters() if id(p) in unused)
        assert record[0] == record[1]
        q.put((rank, "ok", record[0], [(off, p.numel()) for p, off in net.grad_arena_layout()]))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "error: " + repr(e) + "\n" + traceback.format_exc(), None, None))
    finally:
        dist.destroy_process_group()


def _conv_blocks(net):
    from nnuzoo_amd.nets.plain_conv_unet import ConvDropoutNormReLU
    return [m for m in net.modules() if isinstance(m, ConvDropoutNormReLU)]


def test_real_backward_schedule_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), [r[1] for r in res]
    (s0, l0), (s1, l1) = (res[0][2], res[0][3]), (res[1][2], res[1][3])
    assert s0 == s1 and l0 == l1                                   # identical slices and arena layout on every rank
    nbytes = [4 * (hi - lo) for lo, hi in s0]
    assert len(s0) >= 6, s0
    assert all(b >= 12 << 20 for b in nbytes[:-1])                 # every overlapped bucket is at least 12 MB
    ring_ms = 2 * (7 / 8) * nbytes[-1] / 153e9 * 1e3
    assert ring_ms <= 0.28, (nbytes[-1], ring_ms)                  # tail hidden behind enc0.1's weight gradient


def test_recording_reducer_single_process():
    """same schedule, no process group: a recording stand-in for the reducer sees monotonically growing `filled` marks
    - one per conv block and per decoder / encoder stage - ending at the full arena"""
    net = _build_net()

    class Recorder:
        def __init__(self):
            self.marks, self.finished = [], None

        def stage_done_arena(self, arena, filled):
            self.marks.append(filled)

        def finish_arena(self, arena, filled):
            self.finished = filled

    net.grad_reducer = rec = Recorder()
    _one_step(net, fill=1.0)
    total = sum(p.numel() for p in net.parameters())
    # one mark per conv block (22) + one per decoder / encoder stage (5 + 6)
    assert len(rec.marks) == 22 + 5 + 6 and rec.marks == sorted(rec.marks) and rec.finished == total
    # backward starts at the full-resolution decoder stage: its second conv block (32 -> 32, + norm + seg head) comes first;
    # round 3: its data-gradient launch also produces the affine gradients (2 x 32) of the block below (fused norm-backward
    # reductions), which therefore sit in front of the first hand-over mark
    assert rec.marks[0] == 32 * 32 * 27 + 3 * 32 + 2 * 32 + 2 + 2 * 32
