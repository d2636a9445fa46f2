"""oracle/ssnd.py against the reference's own SSND module (tests/golden/ssnd2d.npz, ssnd3d.npz from tools/make_golden.py gen_ssnd:
output, dx and every parameter gradient; the 3-D case pins the reference's reuse of scan order 1).  CPU only."""
import os

import numpy as np
import pytest
import torch

from golden_util import det_fill

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("tag,nd", [("2d", 2), ("3d", 3)])
def test_ssnd_block_equals_the_references(tag, nd):
    from oracle.ssnd import SSND
    z = np.load(os.path.join(G, f"ssnd{tag}.npz"))
    x = torch.tensor(z["x"]).requires_grad_(True)
    m = SSND(nd, x.shape[-1])
    det_fill(m)
    m.eval()
    names = [n for n, _ in m.named_parameters()]
    assert names == [str(n) for n in z["names"]]
    y = m(x)
    ref = torch.tensor(z["y"])
    assert (y - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-7
    grads = torch.autograd.grad(y, [x] + [p for _, p in m.named_parameters()], torch.tensor(z["dy"]))
    for n, g in zip(["dx"] + ["g_" + n for n in names], grads):
        r = torch.tensor(z[n])
        assert (g - r).abs().max().item() <= 1e-4 * r.abs().max().item() + 1e-7, n
