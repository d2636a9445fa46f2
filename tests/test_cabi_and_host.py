"""CPU: (1) libnnuzoo_hip.so loads and exports every symbol declared in include/nnuzoo_hip.h, and the ctypes
signature table covers exactly those symbols (no compute calls without a GPU); (2) host-side plumbing of
BASELINE.json configs[0] (nnUNet 2d, 1x512x512, batch 2, CPU): planner output, deep-supervision scales / weights,
synthetic batch layout, the loud failure of the product path on CPU tensors, and one oracle step as the CPU
reference of that configuration."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "nnuzoo_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long)\s+(nnz_[A-Za-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from nnuzoo_amd import _lib
    lib = _lib.load()
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/nnuzoo_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms, set(_lib.SIGNATURES) ^ set(syms)
    assert lib.nnz_version() == 100
    # chunk-entry states for chunks as short as 64 steps + one state per 16-step sub-block (both kernel generations)
    assert lib.nnz_selective_scan_state_floats(2, 128, 1000) == 2 * 128 * 16 * 16 + 2 * 128 * 63 * 16
    assert lib.nnz_selective_scan_grad_state_floats(2, 128, 1000) == 2 * 128 * 16 * 16


def test_invalid_arguments_are_rejected_without_gpu():
    from nnuzoo_amd import _lib
    lib = _lib.load()
    # NULL pointers / unsupported shapes return EINVAL before any HIP call
    assert lib.nnz_conv_tap_forward(None, None, None, None, None, None) == -22
    assert lib.nnz_window_attention_forward(None, None, None, None, 1, 7, 7, 32, 2, 0, ctypes.c_float(1.0), None) == -22
    assert lib.nnz_selective_scan_forward(*([None] * 10), 1, 4, 8, 16, 64, 1, None) == -22


def test_config0_plumbing_2d_cpu():
    from nnuzoo_amd.synthetic import conv_flops_forward, nnunet_plans, synthetic_batch
    from nnuzoo_amd.training.nnUNetTrainer import nnUNetTrainer
    from oracle.losses import deep_supervision_loss, ds_weights
    from oracle.plain_conv_unet import OraclePlainConvUNet, planner_arch_kwargs
    plans, cfg, dj = nnunet_plans(2, (512, 512), batch_size=2)
    arch = plans["configurations"][cfg]["architecture"]["arch_kwargs"]
    assert cfg == "2d" and arch["n_stages"] == 8
    assert arch["features_per_stage"] == [32, 64, 128, 256, 512, 512, 512, 512]
    assert abs(sum(conv_flops_forward(arch, (512, 512)).values()) / 1e9 - 119.2) < 0.1
    assert plans["configurations"][cfg]["batch_dice"] is True
    tr = nnUNetTrainer(plans, cfg, 0, dj, device=torch.device("cpu"))
    scales = tr._get_deep_supervision_scales()
    assert len(scales) == 7 and scales[0] == [1.0, 1.0] and scales[-1] == [1 / 64, 1 / 64]
    w = ds_weights(7)
    assert w[-1] == 0 and abs(w.sum() - 1) < 1e-12 and abs(w[0] / w[1] - 2) < 1e-12
    batch = synthetic_batch(2, (512, 512), scales, seed=1234)
    assert batch["data"].shape == (2, 1, 512, 512) and batch["data"].dtype == torch.float32
    assert [tuple(t.shape[2:]) for t in batch["target"]] == [(512 >> i, 512 >> i) for i in range(7)]
    assert all(t.dtype == torch.int16 for t in batch["target"])
    # the product network is a HIP schedule (2-D plans run as depth-1 volumes): it builds anywhere, with the
    # reference's state_dict, but a CPU tensor is refused - there is no CPU fallback
    tr.initialize()
    ref_keys = list(OraclePlainConvUNet(1, num_classes=2, **planner_arch_kwargs(2, 8, arch["features_per_stage"]))
                    .state_dict().keys())
    assert list(tr.network.state_dict().keys()) == ref_keys
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tr.network(batch["data"])
    # oracle = the CPU reference of this configuration (reduced to 128^2 here to keep the CPU suite short)
    torch.manual_seed(0)
    net = OraclePlainConvUNet(1, num_classes=2, **planner_arch_kwargs(2, 6, [32, 64, 128, 256, 512, 512]))
    small = synthetic_batch(2, (128, 128), [[1 / 2 ** i] * 2 for i in range(5)], seed=3)
    opt = torch.optim.SGD(net.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        l = deep_supervision_loss(net(small["data"]), small["target"], batch_dice=True)
        l.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 12)
        opt.step()
        losses.append(float(l))
    assert np.isfinite(losses).all() and losses[-1] < losses[0]


def test_product_paths_refuse_cpu_tensors():
    from nnuzoo_amd.selective_scan import selective_scan_fn
    from nnuzoo_amd.training.loss import DC_and_CE_loss, MemoryEfficientSoftDiceLoss
    from nnuzoo_amd.window_attention import window_attention_core
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        selective_scan_fn(torch.zeros(1, 4, 8), torch.zeros(1, 4, 8), torch.zeros(4, 16), torch.zeros(1, 1, 16, 8),
                          torch.zeros(1, 1, 16, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        window_attention_core(torch.zeros(1, 7, 7, 96), torch.zeros(169, 2), torch.zeros(49, 49, dtype=torch.int32), 2, 0, 1.0)
    loss = DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False, 'ddp': False}, {}, weight_ce=1,
                          weight_dice=1, ignore_label=None, dice_class=MemoryEfficientSoftDiceLoss)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        loss(torch.zeros(1, 2, 4, 4), torch.zeros(1, 1, 4, 4, dtype=torch.int16))


def test_row_stride_of_token_major_views():
    """host logic of the in-place readers (gated LayerNorm's z, the fused depthwise conv's x): both halves of the SS2D
    in_proj output are row-strided views; anything that is not such a view is reported so the caller copies"""
    import torch
    from nnuzoo_amd.layer_norm import _row_stride
    xz = torch.zeros(2, 5, 7, 24)
    x, z = xz.chunk(2, dim=-1)
    assert _row_stride(x) == 24 and _row_stride(z) == 24 and _row_stride(xz) == 24
    assert _row_stride(torch.zeros(3, 8)) == 8
    assert _row_stride(xz.permute(0, 2, 1, 3)) is None          # rows no longer uniformly spaced
    assert _row_stride(xz.transpose(-1, -2)) is None            # last dimension strided
    assert _row_stride(xz[:, :, ::2]) is None


def test_projection_weight_gradient_chunking_matches_einsum():
    """the chunked batched-GEMM form used for long sequences (nnuzoo_amd/ss2d_scan.py) is the same contraction"""
    import torch
    from nnuzoo_amd.ss2d_scan import _proj_weight_grad
    g = torch.Generator().manual_seed(0)
    for L in (8192 * 2, 4096, 8192 * 8):
        dP = torch.randn(2, 2, 5, L, generator=g, dtype=torch.float64)
        x2 = torch.randn(2, 2, 3, L, generator=g, dtype=torch.float64)
        ref = torch.einsum("sbcl,sbdl->scd", dP, x2)
        assert torch.allclose(_proj_weight_grad(dP, x2), ref, rtol=1e-10, atol=1e-9)


def test_token_linear_chunking_and_cpu_path():
    """host logic: chunk counts are powers of two that divide the token count and keep >= 4096 tokens per chunk; on CPU
    tensors (e.g. state-dict tooling) the layer is a plain Linear"""
    import torch
    from nnuzoo_amd.token_linear import TokenLinear, _chunks
    assert _chunks(4096) == 1 and _chunks(8192) == 2 and _chunks(524288) == 128 and _chunks(3 * 8192) == 4
    assert _chunks(8191) == 1
    lin = TokenLinear(4, 3)
    x = torch.randn(5, 4)
    assert torch.equal(lin(x), torch.nn.functional.linear(x, lin.weight, lin.bias))
    assert list(lin.state_dict()) == ["weight", "bias"]


def test_ctypes_signatures_match_the_header_declarations():
    """every ctypes signature has the arity and the per-position kind (pointer / int / long / float) of its
    declaration in include/nnuzoo_hip.h - a drifted table would corrupt arguments silently"""
    from nnuzoo_amd import _lib
    txt = open(os.path.join(ROOT, "include", "nnuzoo_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    decls = re.findall(r"\b(?:int|long)\s+(nnz_[A-Za-z0-9_]+)\s*\((.*?)\)\s*;", txt, flags=re.S)
    assert len(decls) == len(_lib.SIGNATURES)

    def kind_of_decl(p):
        p = " ".join(p.split())
        if "*" in p:
            return "ptr"
        if re.search(r"\bfloat\b", p):
            return "float"
        if re.search(r"\bdouble\b", p):
            return "double"
        if re.search(r"\blong\b", p):
            return "long"
        assert re.search(r"\bint\b", p), p
        return "int"

    def kind_of_ctype(t):
        return {ctypes.c_void_p: "ptr", ctypes.c_char_p: "ptr", ctypes.c_int: "int", ctypes.c_long: "long", ctypes.c_float: "float",
                ctypes.c_double: "double"}.get(
            t, "ptr" if isinstance(t, type) and issubclass(t, ctypes._Pointer) else str(t))

    for name, params in decls:
        plist = [p for p in params.split(",") if p.strip() and p.strip() != "void"]
        sig = _lib.SIGNATURES[name]
        assert len(plist) == len(sig), (name, len(plist), len(sig))
        for i, (p, t) in enumerate(zip(plist, sig)):
            assert kind_of_decl(p) == kind_of_ctype(t), (name, i, p.strip(), t)


def test_header_matches_the_extern_c_definitions():
    """include/nnuzoo_hip.h against the `extern "C"` definitions in nnuzoo_amd/csrc/*.hip: same functions, same arity,
    same per-position kind"""
    def strip(txt):
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        return re.sub(r"//[^\n]*", "", txt)

    def kinds(params):
        out = []
        for p in params.split(","):
            p = " ".join(p.split())
            if not p or p == "void":
                continue
            out.append("ptr" if "*" in p else "float" if re.search(r"\bfloat\b", p) else
                       "double" if re.search(r"\bdouble\b", p) else "long" if re.search(r"\blong\b", p) else "int")
        return out

    hdr = strip(open(os.path.join(ROOT, "include", "nnuzoo_hip.h")).read())
    declared = {n: kinds(p) for n, p in re.findall(r"\b(?:int|long)\s+(nnz_[A-Za-z0-9_]+)\s*\((.*?)\)\s*;", hdr, flags=re.S)}
    defined = {}
    csrc = os.path.join(ROOT, "nnuzoo_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith(".hip"):
            src = strip(open(os.path.join(csrc, f)).read())
            for n, p in re.findall(r'extern\s+"C"\s+(?:int|long)\s+(nnz_[A-Za-z0-9_]+)\s*\(([^;{}]*?)\)\s*\{', src, flags=re.S):
                defined[n] = kinds(p)
    assert sorted(declared) == sorted(defined), set(declared) ^ set(defined)
    for n in declared:
        assert declared[n] == defined[n], (n, declared[n], defined[n])


def test_python_call_sites_pass_the_declared_number_of_arguments():
    """static check of every `call("nnz_...", ...)` in the package and the tools: the positional argument count equals
    the ctypes signature's (a missing / extra argument would shift every later one)"""
    import ast
    from nnuzoo_amd import _lib
    checked = 0
    for base in ("nnuzoo_amd", "tools", "tests"):
        for root, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if not f.endswith(".py"):
                    continue
                tree = ast.parse(open(os.path.join(root, f)).read())
                for node in ast.walk(tree):
                    if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id == "call" \
                            and node.args and isinstance(node.args[0], ast.Constant) \
                            and isinstance(node.args[0].value, str) and node.args[0].value.startswith("nnz_"):
                        if any(isinstance(a, ast.Starred) for a in node.args):
                            continue
                        name = node.args[0].value
                        assert name in _lib.SIGNATURES, (f, name)
                        assert len(node.args) - 1 == len(_lib.SIGNATURES[name]), \
                            (os.path.join(root, f), node.lineno, name, len(node.args) - 1, len(_lib.SIGNATURES[name]))
                        checked += 1
    assert checked >= 40


def test_interpolation_matrix_equals_torch_bilinear_on_cpu():
    """common2d._interp_matrix (the separable form behind _upsample_like's backward): Wy x Wx^T reproduces
    F.interpolate(bilinear, align_corners=False) and its transpose is autograd's backward"""
    import torch
    import torch.nn.functional as F
    from nnuzoo_amd.nets.common2d import _interp_matrix
    torch.manual_seed(0)
    for (h, w), (H, W) in (((16, 16), (512, 512)), ((7, 12), (40, 31)), ((33, 20), (33, 20)), ((5, 9), (10, 9))):
        x = torch.randn(2, 3, h, w, requires_grad=True)
        Wy, Wx = _interp_matrix(h, H, "cpu"), _interp_matrix(w, W, "cpu")
        ref = F.interpolate(x, size=(H, W), mode="bilinear", align_corners=False)
        assert torch.allclose(Wy @ x.detach() @ Wx.t(), ref.detach(), atol=1e-5)
        g = torch.randn_like(ref)
        (gr,) = torch.autograd.grad(ref, x, g)
        assert torch.allclose(Wy.t() @ g @ Wx, gr, atol=1e-4)


def test_wgrad_chunk_policy():
    from nnuzoo_amd.token_linear import _wgrad_chunks
    assert _wgrad_chunks(882, 256, 256) == 9            # one 256 x 256 macro tile: cut the 882 tokens (98 per chunk)
    assert 882 % _wgrad_chunks(882, 256, 256) == 0
    assert _wgrad_chunks(35378, 32, 32) >= 32           # tiny result, many tokens
    assert _wgrad_chunks(200, 256, 256) == 1            # too few tokens to cut
    assert _wgrad_chunks(2450, 3072, 3072) == 1         # 144 macro tiles already fill the chip
    assert _wgrad_chunks(524288, 64, 32) == 128         # the tall case of the VSS blocks (unchanged rule)


def test_depthwise_wgrad_routing_and_workspace():
    """host side of csrc/depthwise_wgrad.hip: which depthwise calls common2d sends to it, and its workspace size function
    (one 10-float partial per (channel, batch, band); bands only while a band keeps >= 2048 pixels)"""
    import torch
    from nnuzoo_amd import _lib
    from nnuzoo_amd.nets.common2d import _dw_wgrad_ok
    lib = _lib.load()
    assert lib.nnz_dwconv2d_wgrad_workspace_floats(2, 512, 16, 16) == 512 * 2 * 10            # plenty of planes: one band each
    assert lib.nnz_dwconv2d_wgrad_workspace_floats(2, 32, 512, 512) == 32 * 2 * 32 * 10       # 64 planes: 32 bands of 16 rows
    assert lib.nnz_dwconv2d_wgrad_workspace_floats(1, 1, 8, 8) == 10
    assert lib.nnz_dwconv2d_wgrad_workspace_floats(0, 4, 8, 8) == 0
    assert lib.nnz_dwconv2d_wgrad(None, None, 0, None, None, None, 1, 1, 8, 8, 1, None) == -22
    x, w3, w1, w5 = torch.zeros(1, 4, 8, 8), torch.zeros(4, 1, 3, 3), torch.zeros(4, 1, 1, 1), torch.zeros(4, 1, 5, 5)
    assert _dw_wgrad_ok(x, x, w3, (1, 1), (1, 1), (1, 1))
    assert _dw_wgrad_ok(x, x, w3, (1, 1), (2, 2), (2, 2))            # dilated, padding = dilation
    assert _dw_wgrad_ok(x, x, w1, (1, 1), (0, 0), (1, 1))            # 1x1 depthwise: centre tap
    assert not _dw_wgrad_ok(x, x, w3, (2, 2), (1, 1), (1, 1))        # strided: ATen
    assert not _dw_wgrad_ok(x, x, w3, (1, 1), (0, 0), (1, 1))        # 'valid' padding: output smaller than input
    assert not _dw_wgrad_ok(x, x, w5, (1, 1), (2, 2), (1, 1))        # other kernel sizes
    assert not _dw_wgrad_ok(x.double(), x.double(), w3.double(), (1, 1), (1, 1), (1, 1))
