"""nnuzoo_amd.param_shadow: one multi-tensor fp32 -> fp16 cast of the plain torch convolutions' parameters per autocast step (and
one back for their gradients) instead of a cast launch per parameter and direction - results must be what autocast's own
per-parameter casts give, bit for bit."""
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        from nnuzoo_amd.nets.common2d import Convolution
        self.a = Convolution(2, 3, 16, kernel_size=3)
        self.n = nn.InstanceNorm2d(16)
        self.b = Convolution(2, 16, 16, kernel_size=3, groups=16)        # depthwise: the _Conv2d route
        self.c = nn.Conv2d(16, 8, 1)
        self.lin = nn.Linear(8, 4)                                       # not eligible: stays an autocast cast

    def forward(self, x):
        y = self.c(torch.relu(self.b(self.n(self.a(x)))))
        return self.lin(y.mean((2, 3)))


def test_shadowed_forward_backward_equals_autocast(hip_lib):
    from nnuzoo_amd.param_shadow import ParamShadow, _eligible
    torch.manual_seed(0)
    net = _Net().cuda()
    x = torch.randn(2, 3, 32, 32, device="cuda")
    names, params = _eligible(net)
    assert names == ["a.conv.weight", "a.conv.bias", "b.conv.weight", "b.conv.bias", "c.weight", "c.bias"]
    res = []
    for shadow in (False, True):
        net.zero_grad(set_to_none=True)
        fwd = ParamShadow(net)
        fwd.enabled = shadow
        with torch.autocast("cuda"):
            y = fwd(x)
            loss = (y.float() ** 2).sum()
        loss.backward()
        assert fwd.last_count == (6 if shadow else 0)
        res.append((y.detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters()}))
    assert torch.equal(res[0][0], res[1][0])
    for n in res[0][1]:
        assert res[1][1][n].dtype == torch.float32
        assert torch.equal(res[0][1][n], res[1][1][n]), n
    # outside autocast / without gradients the wrapper is the network
    with torch.no_grad():
        assert torch.equal(ParamShadow(net)(x), net(x))
