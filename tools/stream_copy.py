"""ctypes binding of tools/stream_copy.c — the reference's `io::copy` protocol (8192-byte write() / read() calls,
examples/flate.rs:52,96-97) driven from C over the stream ABI of include/lfx.h.  Bench / test infrastructure."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_SO = os.path.join(_HERE, "liblfx_streamcopy.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "stream_copy.c")
    libdir = os.path.join(_ROOT, "libflate_amd")
    deps = [src, os.path.join(_ROOT, "include", "lfx.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(d) > os.path.getmtime(_SO) for d in deps):
        subprocess.check_call(["gcc", "-O2", "-Wall", "-shared", "-fPIC", "-o", _SO, src, "-L" + libdir, "-llfx",
                               "-Wl,-rpath," + libdir])
    return _SO


def _get():
    global _lib
    if _lib is None:
        from libflate_amd import _ffi
        _ffi.lib()                       # (liblfx.so and the process's one HIP runtime first)
        build()
        _lib = C.CDLL(_SO)
        vp, sz = C.c_void_p, C.c_size_t
        _lib.lfx_sc_encode.argtypes = [vp, C.c_int, vp, vp, sz, sz, vp, sz, C.POINTER(sz), C.POINTER(C.c_double)]
        _lib.lfx_sc_decode.argtypes = [vp, C.c_int, vp, sz, sz, vp, sz, C.POINTER(sz), C.POINTER(C.c_double)]
        _lib.lfx_sc_last_split.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_long)]
        _lib.lfx_sc_memcpy.restype = C.c_double
        _lib.lfx_sc_memcpy.argtypes = [vp, vp, sz, sz]
    return _lib


def encode(ctx, fmt, opts, in_ptr, n, chunk, out_ptr, cap):
    """→ (status, compressed bytes, seconds)"""
    ln, t = C.c_size_t(0), C.c_double(0)
    rc = _get().lfx_sc_encode(ctx.handle, fmt, C.byref(opts) if opts is not None else None, in_ptr, n, chunk, out_ptr, cap,
                              C.byref(ln), C.byref(t))
    return rc, ln.value, t.value


def decode(ctx, fmt, in_ptr, n, chunk, out_ptr, cap):
    """→ (status, bytes produced, seconds)"""
    ln, t = C.c_size_t(0), C.c_double(0)
    rc = _get().lfx_sc_decode(ctx.handle, fmt, in_ptr, n, chunk, out_ptr, cap, C.byref(ln), C.byref(t))
    return rc, ln.value, t.value


def last_split():
    """(seconds in calls that ran GPU work — a batch, a window —, their number) of the last encode / decode"""
    t, k = C.c_double(0), C.c_long(0)
    _get().lfx_sc_last_split(C.byref(t), C.byref(k))
    return t.value, k.value


def memcpy_seconds(dst_ptr, src_ptr, n, chunk):
    return _get().lfx_sc_memcpy(dst_ptr, src_ptr, n, chunk)
