"""Build dronechase_amd/libthreatengage.so in-tree with hipcc for gfx950 (no JIT cache, no pip)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libthreatengage.so")
SOURCES = ["te_env.hip", "te_config.c"]
INCLUDE = os.path.join(PKG, "..", "include")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built (there is no CPU fallback)")


def deps() -> list:
    """Every file the library is compiled from: all of csrc/ (te_env.hip includes te_device.hpp, te_logic.hpp,
    te_stacked.hpp, ...) plus the public headers.  Globbed, so a new header cannot be forgotten."""
    out = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp", ".h", ".c", ".cpp"))]
    out += [os.path.join(INCLUDE, f) for f in sorted(os.listdir(INCLUDE)) if f.endswith(".h")]
    return out


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in deps())


def build_library(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    tag = f".{os.getpid()}"  # unique temporaries: two processes building at once each write their own files and the last
    # os.replace wins with a complete library (never a half-written one)
    obj = os.path.join(CSRC, f"te_config{tag}.o")
    hobj = os.path.join(CSRC, f"te_env{tag}.o")
    tmp = LIB + tag + ".tmp"
    # -fno-slp-vectorize: the SLP vectorizer turns the scalar fp32 physics into v_pk_* pairs; on gfx950 that saves
    # no instructions here (the pairs cost as many v_mov to assemble) but needs 95 instead of 72 VGPRs in the
    # sub-step kernel, i.e. 5 instead of 7 waves per SIMD
    cmds = [["gcc", "-O2", "-fPIC", "-fvisibility=hidden", "-c", os.path.join(CSRC, "te_config.c"), "-o", obj],
            [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fno-gpu-rdc",
             "-fno-slp-vectorize", "-Wall", "-Wno-unused-function", *extra_flags, "-c", os.path.join(CSRC, "te_env.hip"), "-o", hobj],
            [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", hobj, obj, "-o", tmp, "-lm"]]
    try:
        for cmd in cmds:
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
        os.replace(tmp, LIB)
    finally:   # an interrupted or failed build leaves nothing behind (stale objects would travel to the GPU box)
        for f in (obj, hobj, tmp):
            if os.path.exists(f):
                os.remove(f)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
