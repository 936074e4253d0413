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
# one translation unit per kernel family: hipcc compiles them in parallel
SOURCES = ["diral_env.hip", "k_fast64.hip", "k_wide2.hip", "k_wide4.hip", "k_general.hip", "k_observe.hip", "k_large.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc")))

# -ffp-contract=off: the reference (CPython floats) never fuses a*b+c; the bin
# edges, distances and the position wrap must round exactly as it does.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
               "-fno-fast-math", "-fPIC"]
OBJDIR = os.path.join(HERE, "build")


def source_digest() -> str:
    """sha256[:16] over the kernel sources and the C-ABI header, in a fixed order: identifies WHICH kernels a
    library / a profile / a bench line is about on boxes without a .git (profiles/run_profile.sh, bench.py)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    with open(os.path.join(HERE, "..", "include", "diral_env.h"), "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()[:16]


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


def _obj_stale(src: str, obj: str) -> bool:
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS]
    deps.append(os.path.join(HERE, "..", "include", "diral_env.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def variant_objdir(lib: str, extra_flags=()) -> str:
    """Object directory of a build: the product's own, or one per (library name, flag set) for a tuning
    variant - a variant never reuses or overwrites objects compiled with other -D flags."""
    if lib == LIB and not extra_flags:
        return OBJDIR
    import hashlib
    tag = hashlib.sha256(("\0".join([os.path.abspath(lib)] + list(extra_flags))).encode()).hexdigest()[:12]
    return os.path.join(OBJDIR, "variant_" + tag)


def build(force: bool = False, verbose: bool = False, extra_flags=(), lib: str = LIB, objdir: str = None) -> str:
    """Compile every translation unit (in parallel, one hipcc each) and link the .so.
    `extra_flags` / `lib` / `objdir`: tuning variants (-D...) built next to the product, objects in a
    directory of their own (variant_objdir)."""
    if not force and lib == LIB and not extra_flags and not is_stale():
        return lib
    if objdir is None:
        objdir = variant_objdir(lib, extra_flags)
    if lib == LIB and extra_flags:
        raise ValueError("a build with extra flags must not overwrite the product library: pass lib=")
    os.makedirs(objdir, exist_ok=True)
    hipcc = hipcc_path()
    procs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        if not force and not _obj_stale(src, obj):
            continue
        cmd = [hipcc] + HIPCC_FLAGS + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
    failed = [src for src, pr in procs if pr.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed on %s" % ", ".join(failed))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force=True, verbose=True))
