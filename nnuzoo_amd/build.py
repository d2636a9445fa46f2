"""Builds libnnuzoo_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

`python -m nnuzoo_amd.build` or `__graft_entry__.build()`.  The library is a plain shared object with
`extern "C"` entry points (include/nnuzoo_hip.h); nothing links against torch.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnnuzoo_hip.so")
OBJ = os.path.join(HERE, "_obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffast-math", "-fno-finite-math-only",
         "-Wno-unused-result"]


# Files whose results must be bit-identical to IEEE half/float arithmetic (exact division, denormals kept): no fast-math.
STRICT_FP = {"sliding_window.hip", "input_pipeline.hip", "augment.hip"}


def _flags(src: str):
    if src in STRICT_FP:
        return [f for f in FLAGS if f not in ("-ffast-math", "-fno-finite-math-only")] + \
            ["-fno-fast-math", "-fno-gpu-flush-denormals-to-zero"]
    return FLAGS


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(path: str) -> str:
    h = hashlib.sha1()
    for dep in sorted(os.listdir(CSRC)):
        if dep.endswith((".hpp", ".h")) or dep == os.path.basename(path):
            with open(os.path.join(CSRC, dep), "rb") as f:
                h.update(f.read())
    h.update(" ".join(_flags(os.path.basename(path))).encode())
    return h.hexdigest()


def _compile(src: str, force: bool = False, obj_dir: str = None) -> str:
    path = os.path.join(CSRC, src)
    obj = os.path.join(obj_dir or OBJ, src.replace(".hip", ".o"))
    stamp = obj + ".sha1"
    dig = _digest(path)
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj
    cmd = [HIPCC, *_flags(src), "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(dig)
    return obj


# the two smallest sources: always recompiled by build(), so that every call exercises hipcc --offload-arch=gfx950 and the link
# step even when the sha1 stamps of the (git-ignored, shipped) objects say everything is current
ALWAYS = ("device_info.hip", "graph_tools.hip")


def build(verbose: bool = True, jobs: int = 4, force: bool = None) -> str:
    """force=True (or NNZ_BUILD_FORCE=1) recompiles every source; otherwise sources whose digest (file + headers + flags) matches
    the stamp of their object are skipped - except ALWAYS."""
    if force is None:
        force = os.environ.get("NNZ_BUILD_FORCE", "0") == "1"
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s_: _compile(s_, force or s_ in ALWAYS), srcs))
    newest = max(os.path.getmtime(o) for o in objs)
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[nnuzoo_amd.build] {LIB} ({len(srcs)} HIP sources{', all recompiled' if force else ''})")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv[1:] or None)
    sys.exit(0)
