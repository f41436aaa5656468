"""In-tree build of the HIP extension (gfx950 only) with hipcc.  No JIT cache: the .so lives next to
its sources (pynqs_amd/csrc/libpynqs_amd.so) so that it travels with the repository snapshot."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(CSRC, "libpynqs_amd.so")


def hipcc_path() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def source_hash() -> str:
    """sha256 over the native sources (csrc/*.hip, csrc/*.h, include/pynqs_amd.h; names and contents): what a stored profile of a kernel
    (profiles/pmc_*.json, written by tools/pmc_roofline.py) was measured on -- bench.py refuses to quote a profile of another tree."""
    import hashlib

    h = hashlib.sha256()
    files = sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(_HERE, "..", "include", "pynqs_amd.h")]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(_HERE, "..", "include", "pynqs_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile_one(args):
    src, obj, verbose = args
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return obj


def build_native(force: bool = False, verbose: bool = False) -> str:
    """One object per .hip file (recompiled only when it or a header is newer), compiled in parallel, then linked."""
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor

    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max(os.path.getmtime(d) for d in glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(_HERE, "..", "include", "pynqs_amd.h")])
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append((src, obj, verbose))
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(_compile_one, jobs))
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB
