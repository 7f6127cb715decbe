"""CPU: the C-ABI library builds, loads and exports every symbol include/lfx.h declares; compute
calls fail loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ffi():
    import __graft_entry__ as g
    g.build()
    from libflate_amd import _ffi
    return _ffi


def test_exports_match_header(ffi):
    hdr = open(os.path.join(ROOT, "include", "lfx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(lfx_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"lfx_write_cb", "lfx_flush_cb", "lfx_read_cb", "lfx_sink_cb"}
    L = ffi.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(ffi.EXPORTS), declared ^ set(ffi.EXPORTS)
    assert L.lfx_version() == 0x000100


def test_no_cpu_fallback(ffi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    st = C.c_int(0)
    assert not ffi.lib().lfx_ctx_new(0, C.byref(st))
    assert st.value == ffi.E_DEVICE
    import libflate_amd
    with pytest.raises(ffi.DeviceError):
        libflate_amd.Context(0)
    # a NULL context never computes anything
    out_len = C.c_uint64(0)
    assert ffi.lib().lfx_encode_host(None, ffi.GZIP, None, None, b"abc", 3, None, 0, C.byref(out_len)) == ffi.E_DEVICE


def test_product_does_not_touch_oracle():
    # the product tree must not reference the oracle (test infrastructure) in any way
    for base in ("libflate_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".h", ".cpp", ".hip")):
                    txt = open(os.path.join(dp, fn), errors="replace").read()
                    assert "lfo_" not in txt and "oracle" not in txt.lower().replace("oracle/ (test", ""), (dp, fn)


def test_checksum_combine(ffi, oracle):
    import numpy as np
    rng = np.random.default_rng(3)
    d = rng.integers(0, 256, 50000, dtype=np.uint8).tobytes()
    L = ffi.lib()
    for cut in (0, 1, 777, 49999, 50000):
        a, b = d[:cut], d[cut:]
        assert L.lfx_crc32_combine(oracle.crc32(a), oracle.crc32(b), len(b)) == oracle.crc32(d)
        assert L.lfx_adler32_combine(oracle.adler32(a), oracle.adler32(b), len(b)) == oracle.adler32(d)
