"""Build the C-ABI HIP libraries in-tree with hipcc for gfx950:
  libtdx.so       the product library (include/tdx.h)
  libtdx_diag.so  test hooks / timing diagnostics of the GEMM cores (include/tdx_test.h) — never loaded by the product path

The built .so files are git-ignored but travel to the GPU box with the gpurun snapshot, so the box never
needs to compile.  Each translation unit is compiled to its own object (in parallel, only when a source it
depends on is newer), then linked.  `python -m targetdiarization_amd.build [--force]` rebuilds.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libtdx.so")
LIB_DIAG = os.path.join(HERE, "libtdx_diag.so")
SOURCES = ["mf2.hip", "frontend.hip", "paraformer.hip", "eres2net.hip", "mdx.hip", "punc.hip"]
DIAG_SOURCES = ["diag.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def _deps():
    inc = os.path.join(HERE, "..", "include")
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")] + \
           [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]


def _newer(target: str, srcs) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in srcs if os.path.isfile(s))


def _compile(src: str, force: bool, verbose: bool) -> str:
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if force or _newer(obj, [path] + _deps()):
        cmd = [HIPCC] + FLAGS + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return obj


def _link(lib: str, objs, force: bool, verbose: bool) -> None:
    if force or _newer(lib, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, verbose), SOURCES + DIAG_SOURCES))
    _link(LIB, objs[: len(SOURCES)], force, verbose)
    _link(LIB_DIAG, objs[len(SOURCES):], force, verbose)
    return LIB


def check_isa(force: bool = False, verbose: bool = False) -> None:
    """The inline-asm `ds_read_b64_tr_b16` reads (csrc/gemm_h3.hpp h3_tr_read) sit outside the compiler's lgkmcnt bookkeeping: after every
    (re)build the generated gfx950 ISA of mf2.hip is scanned for an instruction that reads their destination registers before the fence
    (tools/asm_tr_hazard.py) — a compiler bump would otherwise change the hazard window silently.  Cached by a stamp file."""
    root = os.path.dirname(HERE)
    src = os.path.join(CSRC, "mf2.hip")
    stamp = os.path.join(OBJ, "isa_checked.stamp")
    tool = os.path.join(root, "tools", "asm_tr_hazard.py")
    if not force and not _newer(stamp, [src, tool] + _deps()):
        return
    asm = os.path.join(OBJ, "mf2_isa.s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "--cuda-device-only", "-S", "-o", asm, src]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC, capture_output=True)
    r = subprocess.run([sys.executable, tool, asm], capture_output=True, text=True)
    os.remove(asm)
    if r.returncode != 0:
        raise RuntimeError("ISA hazard scan failed (an instruction reads an asm ds_read_b64_tr_b16 destination before its fence):\n" + r.stdout + r.stderr)
    open(stamp, "w").write(r.stdout)


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
    check_isa(force="--force" in sys.argv, verbose=True)
