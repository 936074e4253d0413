"""Build recipe for libdiral_env.so (HIP, gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so sits in-tree (diral_amd/libdiral_env.so, git-ignored) and travels
to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdiral_env.so")
SOURCES = ["diral_env.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp"))

# -ffp-contract=off: the reference (CPython floats) never fuses a*b+c; the bin
# edges, distances and the position wrap must round exactly as it does.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
               "-fno-fast-math", "-fPIC", "-shared"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libdiral_env.so cannot be built")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    deps.append(os.path.join(HERE, "..", "include", "diral_env.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    cmd = [hipcc_path()] + HIPCC_FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
