"""Deterministic, RNG-free parameter fill shared by tools/make_golden.py (applied to the REFERENCE modules in the
build container) and the parity tests (applied to the nnuzoo_amd modules): k-th parameter in named_parameters()
order <- scale_k * cos(0.7071 * i + k).  Equal fills + equal outputs => equal wiring AND equal parameter order."""
import torch


def det_fill(module, skip=("A_logs", "Ds", "relative_position_index")):
    with torch.no_grad():
        for k, (name, p) in enumerate(module.named_parameters()):
            if name.split(".")[-1] in skip:
                continue
            n = p.numel()
            i = torch.arange(n, dtype=torch.float64)
            base = torch.cos(0.7071 * i + k).to(torch.float32).view_as(p).to(p.device)
            last = name.split(".")[-1]
            if p.dim() >= 2:
                fan_in = p[0].numel()
                p.copy_(base * (1.0 / fan_in ** 0.5))
            elif "norm" in name or last in ("weight",) and p.dim() == 1:
                p.copy_(1.0 + 0.1 * base if last == "weight" else 0.05 * base)
            else:
                p.copy_(0.05 * base)
        for k, (name, b) in enumerate(module.named_buffers()):
            if name.endswith("running_mean"):
                b.copy_(0.02 * torch.cos(torch.arange(b.numel(), dtype=torch.float32) + k))
            elif name.endswith("running_var"):
                b.copy_(1.0 + 0.1 * torch.cos(torch.arange(b.numel(), dtype=torch.float32) * 0.3 + k) ** 2)


def toy_seg_network(x: torch.Tensor) -> torch.Tensor:
    """Stand-in "network" of the sliding-window fixtures: (B, 1, *spatial) -> fp16 logits (B, 2, *spatial).
    Position dependent and NOT mirror symmetric (rolls), built from operations that are exact in fp32 on inputs that
    are multiples of 1/8 (|x| <= 4) so that CPU and GPU produce bit-identical half outputs."""
    v = x[:, 0].float()
    l0 = v + 0.5 * torch.roll(v, 1, dims=-1)
    l1 = 0.25 * v - torch.roll(v, 1, dims=-2) + 1.0
    return torch.stack([l0, l1], dim=1).to(torch.float16)


def toy_image(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(-32, 33, (1, *shape), generator=g).float() / 8
