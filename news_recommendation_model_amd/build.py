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
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnrm_hotpath.so")
SOURCES = ["pwattn_fwd.hip", "pwattn_fwd_rw.hip", "pwattn_bwd.hip", "pwattn_bwd_rw.hip", "pwattn_bwd_dp.hip", "gemm.hip", "gemm_bf16.hip", "head.hip", "pool_loss.hip",
           "frontend.hip", "capi.hip"]
OBJDIR = os.path.join(HERE, "build")          # per-source objects (git-ignored): only changed sources are recompiled


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


def _provenance_object(hipcc, objdir, extra_flags):
    """A generated translation unit (never part of the digest) that makes the library say what it was built from:
    nrm_source_digest() = sources_digest() at build time, nrm_build_info() = compiler, flags, time."""
    import datetime
    ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout.splitlines()
    ver = next((ln.strip() for ln in ver if "HIP version" in ln or "clang version" in ln), "hipcc")
    info = (f"{ver}; --offload-arch=gfx950 -O3 -std=c++17 -fPIC {' '.join(extra_flags)}".strip()
            + f"; built {datetime.datetime.now(datetime.timezone.utc).strftime('%Y-%m-%dT%H:%M:%SZ')}")
    src = os.path.join(objdir, "provenance.cpp")
    with open(src, "w") as f:
        f.write('extern "C" const char* nrm_source_digest(void) { return "%s"; }\n' % sources_digest())
        f.write('extern "C" const char* nrm_build_info(void) { return "%s"; }\n' % info.replace("\\", "/").replace('"', "'"))
    obj = os.path.join(objdir, "provenance.o")
    subprocess.run([hipcc, "-O2", "-fPIC", "-c", src, "-o", obj], check=True)
    return obj


def library_digest(path=None):
    """nrm_source_digest() of a built library without going through native.load (None: the library is missing or older than
    the entry point)."""
    import ctypes
    path = path or LIB
    if not os.path.exists(path):
        return None
    try:
        lib = ctypes.CDLL(path)
        lib.nrm_source_digest.restype = ctypes.c_char_p
        return lib.nrm_source_digest().decode()
    except (OSError, AttributeError):
        return None


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "nrm_hotpath.h")]
    if any(os.path.getmtime(d) > t for d in deps):
        return True
    return library_digest() != sources_digest()        # (modification times lie after a checkout or a copy: the content decides)


def build(force=False, verbose=True, extra_flags=(), lib=None):
    """Compile every source to an object (in parallel, skipping objects newer than all sources and headers), then link.
    ``extra_flags`` / ``lib``: the timing-diagnostic builds of scripts/_diag (their objects are not cached)."""
    from concurrent.futures import ThreadPoolExecutor
    out = lib or LIB
    if not force and not extra_flags and not needs_build():
        return out
    hipcc = _hipcc()
    # objects of variant builds (scripts/_diag) live outside the package: what ships to the GPU box is the product only
    objdir = OBJDIR if not extra_flags else os.path.join(
        os.environ.get("NRM_DIAG_OBJDIR", os.path.join(tempfile.gettempdir(), "nrm_build_variants")),
        "".join(c if c.isalnum() else "_" for c in " ".join(extra_flags)))
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")] + [os.path.join(HERE, "..", "include", "nrm_hotpath.h")]
    hdr_time = max(os.path.getmtime(h) for h in headers)

    def compile_one(src):
        path, obj = os.path.join(CSRC, src), os.path.join(objdir, src + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_time):
            return obj
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *extra_flags, "-c", path, "-o", obj]
        if verbose:
            print("[nrm build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    objs.append(_provenance_object(hipcc, objdir, extra_flags))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", out + ".tmp"]
    if verbose:
        print("[nrm build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
