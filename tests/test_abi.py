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
