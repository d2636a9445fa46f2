"""`python bench.py --gpus N` must work AS TYPED (the driver's command shape) and under an outer torch.distributed.run.

CPU part (NNZ_BENCH_DRYRUN=1: gloo, kernel launches stubbed by tests/dryrun.py): the parent starts N child ranks before
anything touches the GPU, relays rank 0's single JSON line and exits with the children's status; the ranks run the REAL
trainer set-up (batch split nnUNetTrainer.py:424-429, rank-0 parameter broadcast, reducer attachment) and the real
backward schedule with the bucketed in-place all-reduce of the gradient arena.
GPU part: the real two-rank step - both ranks on the one GPU of the box, gloo carrying the collectives (RCCL refuses two
ranks on one device) - so that the N > 1 code path (process group, reducer hand-over from the launch stream, max-over-ranks
timing, the JSON line) has run with real kernels before an 8-GPU node ever sees it.
Reference: /root/reference/nnunetv2/run/run_training.py:218-232 (mp.spawn, one process per GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=600):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, *args], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines


def test_gpus2_as_typed_self_launches_two_ranks():
    r, lines = _run(["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1"], {"NNZ_BENCH_DRYRUN": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout                       # ONE JSON line, from rank 0
    j = json.loads(lines[0])
    assert j["dryrun"] is True and j["n_gpus"] == 2 and j["rccl_ranks"] == 2
    assert j["allreduce_buckets_per_step"] >= 6            # overlap granularity of the 125 MB arena
    assert j["arena_floats"] == 31195594                   # every parameter of the 3d_fullres net has an arena slice
    assert j["gradients_are_rank_mean"] is True
    assert j["config"]["global_batch"] == 4 and j["scaling"] == "weak"


def test_outer_launcher_form_is_unchanged():
    """the driver's N > 1 form: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N"""
    r, lines = _run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                     "--master-port", "29631", "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"],
                    {"NNZ_BENCH_DRYRUN": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and json.loads(lines[0])["rccl_ranks"] == 2


def test_gpus1_does_not_spawn():
    r, lines = _run(["bench.py", "--steps", "1", "--warmup", "0"], {"NNZ_BENCH_DRYRUN": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["rccl_ranks"] == 0 and j["allreduce_buckets_per_step"] is None


def test_child_failure_is_the_exit_status():
    r, _ = _run(["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--patch", "100"],
                {"NNZ_BENCH_DRYRUN": "1", "NNZ_BENCH_DRYRUN_FAIL": "1"})
    assert r.returncode != 0


@pytest.mark.gpu
@pytest.mark.parametrize("ddp_graph", ["1", "0"])
def test_two_ranks_real_step_one_gpu_gloo(ddp_graph):
    """world size 2 with real kernels: ddp_graph = 1 is the opt-in GraphedDDPStep (hipGraph segments with the all-reduces between
    them, training/graph_step.py), 0 the default eager step whose bucketed all-reduce overlaps the backward schedule"""
    r, lines = _run(["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "2", "--patch", "64", "--no-secondary",
                     "--no-cpu-baseline"], {"NNZ_BENCH_BACKEND": "gloo", "NNZ_BENCH_SHARE_GPU": "1", "NNZ_DDP_GRAPH": ddp_graph},
                    timeout=1200)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    j = json.loads(lines[-1])
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["allreduce_buckets_per_step"] >= 4     # 5-stage net at 64^3
    assert j["config"]["global_batch"] == 4 and j["value"] > 0
    assert 0 < j["final_loss"] < 2.0 or j["final_loss"] < 0     # finite, a Dice+CE value
    # every rank's own clock beside the max the contract uses, and the bytes the step exchanged (every parameter gradient once)
    assert len(j["rank_ms_per_step"]) == 2 and max(j["rank_ms_per_step"]) == pytest.approx(j["ms_per_step"], rel=1e-3)
    assert j["allreduce_bytes_per_step"] > 4 * 1e6
    if ddp_graph == "1":
        # round 4: the N > 1 step replays hipGraph segments with the all-reduces launched between them
        assert j["hip_graph"] is True and j["hip_graph_segments"] >= j["allreduce_buckets_per_step"] >= 4
    else:
        assert j["hip_graph"] is False
