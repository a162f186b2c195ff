"""CPU (cross-compile) checks of the generated gfx950 ISA — no GPU needed.

The K-major operand of the x3 GEMM (csrc/gemm_h3a.hpp) is read with `ds_read_b64_tr_b16` issued through INLINE ASM (the builtin
makes the compiler wait for every outstanding LDS-DMA first: csrc/gemm_h3.hpp, h3_tr_read).  The compiler does not track the
lgkmcnt of asm instructions, so nothing may read their destination registers before the fence; tools/asm_tr_hazard.py scans
the ISA for that.  Also: the steady loops of the hot kernels must be free of scratch (spill) traffic."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def mf2_isa(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which("hipcc")):
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("isa") / "mf2.s")
    src = os.path.join(ROOT, "targetdiarization_amd", "csrc", "mf2.hip")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "--cuda-device-only", "-S", "-o", out, src],
                   check=True, cwd=os.path.dirname(src), capture_output=True)
    return out


def test_asm_transposing_reads_are_fenced(mf2_isa):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_tr_hazard.py"), mf2_isa], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"(\d+) ds_read_b64_tr_b16, 0 hazards", r.stdout)
    assert m and int(m.group(1)) > 100, r.stdout


def test_h3a_kernels_use_the_asm_reads_without_vmcnt0_and_without_spills_in_the_loop(mf2_isa):
    txt = open(mf2_isa).read().split("\n")
    starts = [i for i, l in enumerate(txt) if l.startswith("_ZN3tdx15gemm_h3a_kernel") and "@" in l]
    assert len(starts) == 2            # the attention launch (TWOSEG + gate) and lin_k^T [v|u] (one segment + store)
    for i in starts:
        end = next(j for j in range(i, len(txt)) if txt[j].startswith(".Lfunc_end"))
        body = [x.strip().split(";")[0].strip() for x in txt[i:end]]
        # every run of instructions between two s_barrier that holds 24 MFMAs and LDS-DMA = a steady stage
        stage, stages = [], []
        for y in body:
            if y.startswith("s_barrier"):
                stages.append(stage); stage = []
            elif y:
                stage.append(y)
        steady = [s for s in stages if sum(y.startswith("v_mfma") for y in s) == 24 and any(y.startswith("global_load_lds") for y in s)]
        assert steady, "no steady stage found"
        for s in steady:
            assert sum(y.startswith("ds_read_b64_tr_b16") for y in s) == 16
            mf = [j for j, y in enumerate(s) if y.startswith("v_mfma")]
            assert not any(y.startswith("scratch_") for y in s[mf[0]:mf[-1]]), "spill traffic between the MFMAs of a stage"
            assert not any(y.startswith("s_waitcnt vmcnt(0)") for y in s[mf[0]:mf[-1]]), "the ring is serialised: vmcnt(0) between the MFMAs of a stage"
