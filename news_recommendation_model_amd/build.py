"""Build libnrm_hotpath.so (the C-ABI library of include/nrm_hotpath.h) for gfx950 with hipcc.

    python -m news_recommendation_model_amd.build          # rebuild if sources are newer
hipcc cross-compiles without a GPU; the .so is written next to this file (in-tree, git-ignored) so it
travels with the repository snapshot to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnrm_hotpath.so")
SOURCES = ["pwattn_fwd.hip", "pwattn_fwd_rw.hip", "pwattn_bwd.hip", "gemm.hip", "head.hip", "pool_loss.hip", "frontend.hip", "capi.hip"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def sources_digest():
    """sha256 over the kernel sources (csrc/* and the C header), in name order: what a profile under profiles/ is
    stamped with (scripts/pmc_summary.py) and what bench.py compares before quoting a stored PMC figure -- the GPU box
    has no .git, so the stamp is over file contents, not a commit id."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(HERE, "..", "include", "nrm_hotpath.h")]
    for path in files:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "nrm_hotpath.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB + ".tmp"]
    if verbose:
        print("[nrm build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
