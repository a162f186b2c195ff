"""CPU tests of the boundary: libtdx.so loads and exports every symbol include/tdx.h declares
(no compute calls without a GPU), argument validation works without touching a device, and
the product path fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

from targetdiarization_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from targetdiarization_amd.build import build_lib
    build_lib()
    return _lib.lib()


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "tdx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(tdx_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in tdx.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert lib.tdx_version().startswith(b"tdx ")


def _exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("tdx_")}


def _declared(header):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return set(re.findall(r"\b(tdx_[a-z0-9_]+)\s*\(", hdr))


def test_exports_are_exactly_the_headers(lib):
    """libtdx.so exports nothing but what include/tdx.h declares (test hooks and timing diagnostics live in
    libtdx_diag.so / include/tdx_test.h), and vice versa for both libraries."""
    assert _exported(_lib.LIB_PATH) == _declared("tdx.h")
    assert _exported(_lib.DIAG_PATH) == _declared("tdx_test.h")
    assert set(_lib.DIAG_SIGNATURES) == _declared("tdx_test.h")
    assert set(_lib.SIGNATURES) == _declared("tdx.h")


def test_corrupt_blobs_are_rejected(lib, sd2):
    """TDXW header fields are untrusted: counts, dim products and offsets that would wrap or point outside the
    buffer must give TDX_E_BLOB, never a crash or an out-of-bounds pointer."""
    import struct
    from targetdiarization_amd.weights import pack_blob
    cfg = _lib.Mf2Config(num_blocks=2, channels=512, kernel_size=16, num_spks=2, group_size=256)
    h = C.c_void_p()

    def create(blob):
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        return lib.tdx_mf2_create(C.byref(cfg), buf, len(blob), 0, C.byref(h))

    def entry(name, dims, off):
        return struct.pack("<H", len(name)) + name + struct.pack("<B", len(dims)) + b"".join(struct.pack("<I", d) for d in dims) + struct.pack("<Q", off)

    def blob_of(entries, n=None, data=b"\0" * 256):
        head = b"TDXW0001" + struct.pack("<I", len(entries) if n is None else n) + b"".join(entries)
        head += b"\0" * ((-len(head)) % 64)
        return head + data

    assert create(blob_of([], n=0xFFFFFFFF)) == 2                                     # entry count beyond the buffer
    assert create(blob_of([entry(b"a", [0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF], 0)])) == 2  # numel product overflows
    assert create(blob_of([entry(b"a", [16], 0xFFFFFFFFFFFFFFF0)])) == 2              # offset + size wraps in u64
    assert create(blob_of([entry(b"a", [16], 2)])) == 2                               # misaligned offset
    assert create(blob_of([entry(b"a", [1 << 20], 0)])) == 2                          # data beyond the buffer
    assert create(blob_of([entry(b"a", [4], 0), entry(b"a", [4], 64)])) == 2          # duplicate name
    assert create(b"TDXW0001" + struct.pack("<I", 1) + b"\x05\x00ab") == 2            # truncated entry
    good = pack_blob(sd2)
    assert create(good[: len(good) // 2]) == 2                                        # truncated data section


def test_unexpected_tensor_is_rejected(lib, sd2):
    """strict both ways, like load_state_dict(strict=True) (base_model.py:63)"""
    from targetdiarization_amd.weights import pack_blob
    sd = dict(sd2)
    sd["mask_net.not_a_parameter"] = torch.zeros(3)
    blob = pack_blob(sd)
    cfg = _lib.Mf2Config(num_blocks=2, channels=512, kernel_size=16, num_spks=2, group_size=256)
    h = C.c_void_p()
    buf = (C.c_char * len(blob)).from_buffer_copy(blob)
    assert lib.tdx_mf2_create(C.byref(cfg), buf, len(blob), 0, C.byref(h)) == 2
    assert b"mask_net.not_a_parameter" in lib.tdx_last_error()


def test_argument_validation_without_gpu(lib):
    assert lib.tdx_mf2_create(None, None, 0, 0, None) == 1
    assert b"null" in lib.tdx_last_error()
    cfg = _lib.Mf2Config(num_blocks=2, channels=256, kernel_size=16, num_spks=2, group_size=256)
    h = C.c_void_p()
    junk = (C.c_char * 64)()
    assert lib.tdx_mf2_create(C.byref(cfg), junk, 64, 0, C.byref(h)) == 1       # unsupported config
    cfg.channels = 512
    assert lib.tdx_mf2_create(C.byref(cfg), junk, 64, 0, C.byref(h)) == 2       # malformed blob
    assert lib.tdx_mf2_workspace_bytes(None, 1, 16000) == 0
    assert lib.tdx_cal_attention_workspace_bytes(1, 100, 65) == 0
    assert lib.tdx_cal_attention_workspace_bytes(1, 100, 64) > 0
    assert lib.tdx_linear(None, None, None, 1, 128, 32, None, None) == 1


def test_missing_tensor_is_reported(lib, sd2):
    from targetdiarization_amd.weights import pack_blob
    sd = dict(sd2)
    sd.pop("dec.weight")
    blob = pack_blob(sd)
    cfg = _lib.Mf2Config(num_blocks=2, channels=512, kernel_size=16, num_spks=2, group_size=256)
    h = C.c_void_p()
    buf = (C.c_char * len(blob)).from_buffer_copy(blob)
    assert lib.tdx_mf2_create(C.byref(cfg), buf, len(blob), 0, C.byref(h)) == 2
    assert b"dec.weight" in lib.tdx_last_error()


def test_no_cpu_fallback(sd2):
    from targetdiarization_amd.separator import MossFormer2Separator
    with pytest.raises(_lib.TdxError):
        MossFormer2Separator(sd2, device="cpu")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "targetdiarization_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith(".py"):
                src = open(os.path.join(dp, fn)).read()
                assert "import oracle" not in src and "from oracle" not in src, fn
