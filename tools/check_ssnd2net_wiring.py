"""Build-container check (needs /root/reference): whole-network wiring of nnuzoo_amd's SSND2NetP / SSND2Net against the
reference's classes (nets/ssnd2net.py under tools/ref_shim.py).  Both networks get the same RNG-free parameters and
the SAME selective-scan implementation (oracle/selective_scan.py on CPU), so any difference in the outputs is a wiring
difference.  Expected: max |diff| == 0.0 for every deep-supervision output (2-D 96^2 and 3-D 24^3).

Why this is a tool and not a fixture: with the networks' ~100 normalisation layers in sequence, two CORRECT scan
implementations that differ by 1e-6 end up 20 % apart at the output, so whole-net outputs cannot pin the HIP path;
tests/test_ssnd2net.py pins it at MU-stage depth instead."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE, os.path.join(os.path.dirname(HERE), "tests")]

import ref_shim  # noqa: E402

ref_shim.install()
from golden_util import det_fill  # noqa: E402
from nnunetv2.nets import ssnd2net as R  # noqa: E402
from nnuzoo_amd.nets import ssnd2net as M  # noqa: E402
from oracle.selective_scan import selective_scan_torch  # noqa: E402


def scan(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False, return_last_state=False):
    return selective_scan_torch(u, delta, A, B, C, D, delta_bias, delta_softplus)


def main():
    worst = 0.0
    for cls in sys.argv[1:] or ["SSND2NetP"]:
        for patch in [(96, 96), (24, 24, 24)]:
            kw = dict(spatial_dims=len(patch), factorization_type="cross-scan", in_ch=1, out_ch=2, deep_supervision=True,
                      input_patch_size=list(patch))
            r, m = getattr(R, cls)(**kw), getattr(M, cls)(**kw)
            for net in (r, m):
                det_fill(net)
                net.eval()
                for mod in net.modules():
                    if hasattr(mod, "selective_scan"):
                        mod.selective_scan = scan
            i = torch.arange(int(np.prod(patch)), dtype=torch.float64)
            x = torch.cos(0.173 * i + 0.3).float().reshape(1, 1, *patch)
            with torch.no_grad():
                yr, ym = r(x), m(x)
            d = [float((a - b).abs().max()) for a, b in zip(yr, ym)]
            worst = max(worst, *d)
            print(cls, patch, "max |diff| per output:", d)
    print("OK" if worst == 0.0 else f"MISMATCH {worst}")
    return 0 if worst == 0.0 else 1


if __name__ == "__main__":
    sys.exit(main())
