"""Makes tests/golden/mamba2_mixer.npz: inputs, parameters, outputs and gradients of HuggingFace transformers'
Mamba2Mixer (pure-torch chunked SSD path on CPU) for the two LightMamba2Net mixer shapes used in the tests.  Run in the
build container (needs `transformers`, no reference import): python tools/make_mamba2_golden.py
The LightMamba2Net call site (/root/reference/nnunetv2/nets/light_mamba2net.py:51-74) fixes d_state=16, d_conv=4,
expand=2, headdim=get_nheaddim(d_model, 2)."""
import os

import numpy as np
import torch
from transformers.models.mamba2.configuration_mamba2 import Mamba2Config
from transformers.models.mamba2.modeling_mamba2 import Mamba2Mixer

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "mamba2_mixer.npz")
NAMES = ["in_proj.weight", "conv1d.weight", "conv1d.bias", "dt_bias", "A_log", "D", "norm.weight", "out_proj.weight"]


def case(tag, d_model, headdim, B, L, seed, chunk):
    torch.manual_seed(seed)
    d_inner = 2 * d_model
    cfg = Mamba2Config(num_heads=d_inner // headdim, head_dim=headdim, hidden_size=d_model, state_size=16, expand=2,
                       conv_kernel=4, n_groups=1, use_bias=False, use_conv_bias=True, chunk_size=chunk,
                       layer_norm_epsilon=1e-5, num_hidden_layers=1)
    mix = Mamba2Mixer(cfg, layer_idx=0).double()
    with torch.no_grad():     # away from the init's special values so every parameter matters
        mix.D.copy_(torch.rand_like(mix.D) + 0.5)
        mix.norm.weight.copy_(torch.rand_like(mix.norm.weight) + 0.5)
        mix.conv1d.bias.copy_(torch.randn_like(mix.conv1d.bias) * 0.1)
    u = torch.randn(B, L, d_model, dtype=torch.float64, requires_grad=True)
    out = mix(u)
    gout = torch.randn_like(out)
    (out * gout).sum().backward()
    d = {f"{tag}.u": u.detach().numpy(), f"{tag}.out": out.detach().numpy(), f"{tag}.gout": gout.numpy(),
         f"{tag}.du": u.grad.numpy(), f"{tag}.headdim": np.array(headdim)}
    sd = dict(mix.named_parameters())
    for n in NAMES:
        d[f"{tag}.p.{n}"] = sd[n].detach().numpy()
        d[f"{tag}.g.{n}"] = sd[n].grad.numpy()
    return d


if __name__ == "__main__":
    data = {}
    data.update(case("a", d_model=16, headdim=2, B=2, L=70, seed=1, chunk=32))     # stage-1 width, ragged chunks
    data.update(case("b", d_model=64, headdim=8, B=1, L=130, seed=2, chunk=64))
    np.savez_compressed(OUT, **{k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in data.items()})
    print(OUT, os.path.getsize(OUT))
