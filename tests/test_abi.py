"""CPU: the C-ABI library builds, loads and exports every symbol include/lfx.h declares; compute
calls fail loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ffi():
    import __graft_entry__ as g
    g.build()
    from libflate_amd import _ffi
    return _ffi


def _declared(name):
    hdr = open(os.path.join(ROOT, "include", name)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return set(re.findall(r"\b(lfx_[a-z0-9_]+)\s*\(", hdr)) - {"lfx_write_cb", "lfx_flush_cb", "lfx_read_cb", "lfx_sink_cb"}


def test_exports_match_header(ffi):
    declared = _declared("lfx.h")
    L = ffi.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(ffi.EXPORTS), declared ^ set(ffi.EXPORTS)
    # the library exports nothing beyond the boundary and the four host-only test hooks (include/lfx_testhooks.h)
    hooks = _declared("lfx_testhooks.h")
    assert hooks and not (hooks & declared) and all(h.startswith("lfx_debug_") for h in hooks)
    out = subprocess.run(["nm", "-D", "--defined-only", ffi.SO_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if l.split()[-1].startswith("lfx_")}
    assert exported == declared | hooks, exported ^ (declared | hooks)
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


# ---- the Rust shim crate (rust/libflate-amd): cannot be compiled here (no Rust toolchain), so the ABI it binds is
# exercised from C: tests/c/shim_abi.c drives every extern the crate declares.
def _shim_binary():
    import subprocess
    src = os.path.join(ROOT, "tests", "c", "shim_abi.c")
    exe = os.path.join(ROOT, "tests", "c", "shim_abi")
    libdir = os.path.join(ROOT, "libflate_amd")
    if not os.path.exists(exe) or os.path.getmtime(src) > os.path.getmtime(exe):
        subprocess.check_call(["gcc", "-O1", "-Wall", "-o", exe, src, "-L" + libdir, "-llfx", "-ldl", "-Wl,-rpath," + libdir])
    return exe


def _shim_env():
    env = dict(os.environ)
    # same HIP runtime as the Python tests use (torch's bundled copy), found through the loader path
    try:
        import torch
        tl = os.path.join(os.path.dirname(torch.__file__), "lib")
        env["LD_LIBRARY_PATH"] = tl + ":" + env.get("LD_LIBRARY_PATH", "") + ":/opt/rocm/lib"
    except ImportError:
        env["LD_LIBRARY_PATH"] = env.get("LD_LIBRARY_PATH", "") + ":/opt/rocm/lib"
    return env


def test_rust_shim_binds_declared_abi(ffi):
    rs = open(os.path.join(ROOT, "rust", "libflate-amd", "src", "ffi.rs")).read()
    bound = set(re.findall(r"pub fn (lfx_[a-z0-9_]+)\s*\(", rs))
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "lfx.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(lfx_[a-z0-9_]+)\s*\(", hdr))
    assert bound and bound <= declared, bound - declared
    csrc = open(os.path.join(ROOT, "tests", "c", "shim_abi.c")).read()
    driven = set(re.findall(r"\b(lfx_[a-z0-9_]+)\s*\(", csrc))
    assert bound <= driven, sorted(bound - driven)        # every extern of the crate is exercised from C
    # struct layouts the crate mirrors by hand
    assert C.sizeof(ffi.EncodeOpts) == 72 and C.sizeof(ffi.Header) == 56
    for field, off in (("block_size", 0), ("window_size", 20), ("mtime", 32), ("os", 36), ("extra", 40), ("extra_len", 48),
                       ("filename", 56), ("comment", 64)):
        assert getattr(ffi.EncodeOpts, field).offset == off, field


def test_shim_abi_program_without_gpu(ffi):
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: tests/test_gpu_round2.py runs the vectors")
    out = subprocess.run([_shim_binary()], env=_shim_env(), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "no device" in out.stdout, out.stderr
