"""TEST INFRASTRUCTURE ONLY - CPU restatement of the Mamba2 ("SSD") mixer LightMamba2Net binds
(/root/reference/nnunetv2/nets/light_mamba2net.py:17, :51-89 -> mamba_ssm.modules.mamba2.Mamba2, third-party, absent from
/root/reference and unpinned there).  Written as the literal per-token recurrence of the published block:

    [z | xBC | dt] = in_proj(u);  xBC = silu(causal depthwise conv1d(xBC));  x, B, C = split(xBC)
    dt = softplus(dt + dt_bias);  H_t = exp(dt_t A_h) H_{t-1} + dt_t x_t (x) B_t;  y_t = H_t C_t + D_h x_t
    out = out_proj( rmsnorm(y * silu(z)) * w )

Pinned by tests/golden/mamba2_mixer.npz: outputs AND parameter gradients of HuggingFace transformers' `Mamba2Mixer`
(its chunked pure-torch path - an independent implementation of the same block), generated in the build container by
tools/make_mamba2_golden.py.  Only tests may import this file.
"""
import torch
import torch.nn.functional as F


def mamba2_mixer(u, p, headdim, d_state=16, d_conv=4, eps=1e-5):
    """u (B, L, d_model) float64/32 tensor; p: dict with in_proj.weight, conv1d.weight (C, 1, W), conv1d.bias, dt_bias,
    A_log, D, norm.weight, out_proj.weight.  Differentiable (torch ops only, time loop)."""
    Bt, L, _ = u.shape
    nheads = p["A_log"].shape[0]
    ds = nheads * headdim
    zxbcdt = F.linear(u, p["in_proj.weight"])
    z, xBC, dt = zxbcdt[..., :ds], zxbcdt[..., ds:2 * ds + 2 * d_state], zxbcdt[..., 2 * ds + 2 * d_state:]
    w = p["conv1d.weight"]
    xBC = F.conv1d(xBC.transpose(1, 2), w, p["conv1d.bias"], padding=d_conv - 1, groups=w.shape[0])[..., :L]
    xBC = F.silu(xBC).transpose(1, 2)
    x = xBC[..., :ds].reshape(Bt, L, nheads, headdim)
    Bm, Cm = xBC[..., ds:ds + d_state], xBC[..., ds + d_state:]
    dt = F.softplus(dt + p["dt_bias"])                                   # (B, L, H)
    A = -torch.exp(p["A_log"])                                           # (H)
    H = u.new_zeros(Bt, nheads, headdim, d_state)
    ys = []
    for t in range(L):
        a = torch.exp(dt[:, t] * A)                                      # (B, H)
        H = a[..., None, None] * H + (dt[:, t, :, None] * x[:, t])[..., None] * Bm[:, t, None, None, :]
        ys.append((H * Cm[:, t, None, None, :]).sum(-1) + p["D"][None, :, None] * x[:, t])
    y = torch.stack(ys, 1).reshape(Bt, L, ds)
    g = y * F.silu(z)
    g = g * torch.rsqrt(g.pow(2).mean(-1, keepdim=True) + eps) * p["norm.weight"]
    return F.linear(g, p["out_proj.weight"])
