"""hipGraph safety of the C-ABI launchers (hipGraph): a captured launch sequence must replay correctly
after the process has made further allocations.  Regression test for the round-1 finding that a captured hipMemsetAsync
node replays with a corrupted fill pattern (tools/probes/graph_memset_repro.py): every zeroing in the library is a kernel."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _junk():
    j = [torch.full((1 + 37 * i,), float("nan"), device="cuda") for i in range(2000)]
    torch.cuda.synchronize()
    del j


def _capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        fn()
    return g


def test_layer_norm_backward_replays_after_allocations(hip_lib):
    from nnuzoo_amd._lib import call, ptr, stream_ptr
    torch.manual_seed(0)
    R, C = 1152, 32
    x, dy, w = torch.randn(R, C, device="cuda"), torch.randn(R, C, device="cuda"), torch.ones(C, device="cuda")
    mean, rstd = x.mean(1).contiguous(), (x.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    dx, dwb = torch.empty_like(x), torch.empty(2, C, device="cuda")

    def run():
        call("nnz_layer_norm_backward", ptr(x), 0, ptr(w), ptr(mean), ptr(rstd), ptr(dy), 0, ptr(dx), ptr(dwb[0]),
             ptr(dwb[1]), 0, R, C, stream_ptr())

    run()
    torch.cuda.synchronize()
    ref_dx, ref_dwb = dx.clone(), dwb.clone()
    assert torch.allclose(ref_dwb[1], dy.sum(0), rtol=1e-4, atol=1e-3)
    g = _capture(run)
    for _ in range(3):
        dx.fill_(777.0)
        dwb.fill_(777.0)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(dx, ref_dx)
        assert torch.allclose(dwb, ref_dwb, rtol=1e-4, atol=1e-3), (dwb - ref_dwb).abs().max().item()
        _junk()


def test_ss2d_block_forward_backward_replays_after_allocations(hip_lib):
    """the fused SS2D block (dwconv, cross-scan, merge, gated norm: memset-free accumulators everywhere) under capture"""
    from nnuzoo_amd.nets.m2net import VSSBlock
    torch.manual_seed(1)
    blk = VSSBlock(hidden_dim=16, drop_path=0.0).cuda()
    x = torch.randn(2, 24, 40, 16, device="cuda", requires_grad=True)
    dy = torch.randn(2, 24, 40, 16, device="cuda")
    params = list(blk.parameters())
    state = {}

    def run():
        for p in params:
            p.grad = None
        x.grad = None
        with torch.autocast("cuda", dtype=torch.float16):
            state["y"] = blk(x)
        state["y"].float().backward(dy)

    g = _capture(run)
    g.replay()
    torch.cuda.synchronize()
    ref = [state["y"].detach().float().clone(), x.grad.clone()] + [p.grad.clone() for p in params]
    assert all(bool(torch.isfinite(t).all()) for t in ref)
    for _ in range(2):
        _junk()
        g.replay()
        torch.cuda.synchronize()
        now = [state["y"].detach().float(), x.grad] + [p.grad for p in params]
        for a, b in zip(now, ref):
            assert torch.allclose(a, b, rtol=2e-3, atol=2e-3 * b.abs().max().item() + 1e-7), (a - b).abs().max().item()


def test_memset_nodes_are_rewritten_and_replay_after_allocations(hip_lib):
    """ATen ops that zero through hipMemsetAsync (Tensor.zero_ on a contiguous tensor, the semaphores of a multi-block
    sum) captured next to our kernels: the rewriting pass (csrc/graph_tools.hip) replaces their memset nodes by fill-kernel
    nodes; the replayed graph has no memset node and reproduces the eager results after the process allocated more."""
    import ctypes as C
    from nnuzoo_amd._lib import call
    from nnuzoo_amd.training.graph_step import capture_memset_free
    torch.manual_seed(0)
    z = torch.empty(4099, device="cuda")
    zb = torch.empty(1000, dtype=torch.uint8, device="cuda")
    big = torch.randn(64, 1 << 16, device="cuda")            # column sums: a multi-block (global) reduction
    out = torch.empty(64, device="cuda")
    tot = torch.empty((), device="cuda")

    def run():
        z.zero_()
        zb.zero_()
        torch.sum(big, dim=1, out=out)
        torch.sum(big, dim=(0, 1), out=tot)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ref_out, ref_tot = out.clone(), tot.clone()
    g, replaced = capture_memset_free(run, side)
    assert replaced >= 1, "expected ATen to emit at least one memset node for these ops"
    counts = (C.c_int * 16)()
    call("nnz_graph_node_census", C.c_void_p(int(g.raw_cuda_graph())), counts, 16)
    assert counts[2] == 0 and counts[0] >= replaced            # hipGraphNodeTypeMemset = 2, Kernel = 0
    for _ in range(3):
        z.fill_(7.0); zb.fill_(9); out.fill_(float("nan")); tot.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        assert float(z.abs().max()) == 0.0 and int(zb.max()) == 0
        assert torch.allclose(out, ref_out, rtol=1e-5, atol=1e-3) and torch.allclose(tot, ref_tot, rtol=1e-5, atol=1e-2)
        _junk()


def test_graph_and_eager_steps_alternate(hip_lib):
    """the captured backward writes into the .grad tensors that existed at capture time.  An eager step in
    between (optimizer.zero_grad(set_to_none=True) + fresh gradients) must not leave later replays writing into orphaned
    buffers: after every graph step p.grad IS the static tensor and holds this step's gradient, and training moves."""
    from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
    from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerM2NetP
    plans, cfg, dj = nnunet_plans(2, (64, 64), batch_size=2)
    torch.manual_seed(0)
    tr = nnUNetTrainerM2NetP(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    assert tr.use_hip_graph
    tr.grad_scaler = torch.amp.GradScaler("cuda", init_scale=256.0)   # no fp16 overflow / skipped steps in this short run
    b = synthetic_batch(2, (64, 64), tr._get_deep_supervision_scales(), seed=5)
    b = {"data": b["data"].cuda(), "target": [t.cuda() for t in b["target"]]}
    losses = [float(tr.train_step(b)["loss"]) for _ in range(2)]             # graph (captures on the first call)
    static = dict((id(p), g) for p, g in tr._graphed._static_grads)
    assert len(static) > 100
    for rnd in range(3):
        tr.use_hip_graph = False
        losses.append(float(tr.train_step(b)["loss"]))                       # eager: zero_grad(set_to_none) + new grads
        moved = [p for p in tr.network.parameters() if p.grad is not None and p.grad is not static.get(id(p))]
        assert len(moved) > 100                                              # the eager step really swapped the tensors
        tr.use_hip_graph = True
        tr.optimizer.zero_grad(set_to_none=True)                             # and a user-side zero_grad on top
        for p, g in tr._graphed._static_grads:
            g.fill_(7777.0)                                                  # a replay that did not write would show
        before = [p.detach().clone() for p in tr.network.parameters()]
        losses.append(float(tr.train_step(b)["loss"]))
        for p in tr.network.parameters():
            if id(p) in static:
                assert p.grad is static[id(p)] and bool(torch.isfinite(p.grad).all())
                assert not bool((p.grad == 7777.0).all())
        changed = sum(int(not torch.equal(a, p.detach())) for a, p in zip(before, tr.network.parameters()))
        assert changed > 100                                                 # the optimizer saw the replayed gradients
    assert all(l == l for l in losses) and losses[-1] < losses[0]
    assert torch.backends.cudnn.enabled                                      # no trainer leaves the library switched off


def test_no_miopen_scope_is_restored(hip_lib):
    """MambaND2Net / UNETR2Net / LightMamba2Net disable the MIOpen path INSIDE their steps only"""
    from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerMambaND2Net, nnUNetTrainerM2Net
    from nnuzoo_amd.synthetic import nnunet_plans
    plans, cfg, dj = nnunet_plans(2, (64, 64), batch_size=2)
    tr = nnUNetTrainerMambaND2Net(plans, cfg, 0, dj, device=torch.device("cuda"))
    assert tr._no_miopen and not nnUNetTrainerM2Net._no_miopen
    assert torch.backends.cudnn.enabled
    with tr._library_scope():
        assert not torch.backends.cudnn.enabled
    assert torch.backends.cudnn.enabled
