"""Build libtdx.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

The built .so is git-ignored but travels to the GPU box with the gpurun snapshot, so the
box never needs to compile.  `python -m targetdiarization_amd.build` rebuilds.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtdx.so")
SOURCES = ["mf2.hip", "frontend.hip", "paraformer.hip", "eres2net.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "tdx.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
