"""The build really builds: hipcc cross-compiles a source for gfx950 from nothing into a scratch directory (no stamp, no shipped
object involved) and the result is a gfx950 code object with the expected kernels.  CPU only (no GPU needed to compile)."""
import os
import subprocess

from nnuzoo_amd import build as B


def test_hipcc_compiles_a_source_from_clean(tmp_path):
    obj = B._compile("graph_tools.hip", force=True, obj_dir=str(tmp_path))
    assert os.path.getsize(obj) > 1000 and os.path.exists(obj + ".sha1")
    # the fat object embeds a gfx950 code object holding the fill kernel of the memset-rewriting pass
    out = subprocess.run(["strings", "-a", obj], capture_output=True, text=True).stdout
    assert "gfx950" in out and "graph_fill_kernel" in out


def test_stamp_follows_sources_headers_and_flags(tmp_path, monkeypatch):
    d0 = B._digest(os.path.join(B.CSRC, "device_info.hip"))
    monkeypatch.setattr(B, "FLAGS", B.FLAGS + ["-DNNZ_TEST_FLAG"])
    assert B._digest(os.path.join(B.CSRC, "device_info.hip")) != d0
    assert set(B.ALWAYS) <= set(B._sources())
