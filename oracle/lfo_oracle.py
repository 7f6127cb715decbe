"""ctypes binding of the C oracle (oracle/liblfo_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under libflate_amd/ may import this.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblfo_oracle.so")

DEFLATE, ZLIB, GZIP = 0, 1, 2
OK, INVALID_DATA, UNEXPECTED_EOF = 0, 1, 2
LZ77_DEFAULT, LZ77_NOCOMPRESSION = 0, 1


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("lfo_core.c", "lfo_deflate.c", "lfo.h")]
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class Buf(C.Structure):
    _fields_ = [("p", C.POINTER(C.c_uint8)), ("n", C.c_size_t), ("cap", C.c_size_t)]


class Opts(C.Structure):
    _fields_ = [
        ("block_size", C.c_size_t), ("dynamic_huffman", C.c_int), ("no_compression", C.c_int),
        ("lz77_kind", C.c_int), ("window_size", C.c_uint32), ("max_length", C.c_uint32),
        ("zlib_sync_flush", C.c_int), ("mtime", C.c_uint32), ("os", C.c_uint8),
        ("is_text", C.c_int), ("hcrc", C.c_int), ("extra", C.c_char_p), ("extra_len", C.c_size_t),
        ("filename", C.c_char_p), ("comment", C.c_char_p),
        ("custom_cb", C.c_void_p), ("custom_user", C.c_void_p), ("custom_level", C.c_int),
    ]


# the generic `E: Lz77Encode` (lfo.h: custom_cb): size_t cb(user, p, n, const uint32_t **codes); p == NULL is E::flush
CUSTOM_CB = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.POINTER(C.c_uint32)))
LZ77_DEFAULT, LZ77_NOCOMPRESSION, LZ77_CUSTOM = 0, 1, 2


def custom_lz77(obj):
    """obj: any object with the Lz77Encode methods encode(buf, sink) / flush(sink) / compression_level() / window_size()
    whose sink receives ("Literal", b) / ("Pointer", length, distance) tuples → keyword options for Encoder / encode."""
    keep = {}

    def cb(_user, p, n, out):
        sink = []
        if p:
            obj.encode(C.string_at(p, n), sink)
        else:
            obj.flush(sink)
        words = (C.c_uint32 * max(len(sink), 1))(*[(c[1] << 16) if c[0] == "Literal" else (c[1] << 16) | c[2] for c in sink])
        keep["words"] = words
        out[0] = C.cast(words, C.POINTER(C.c_uint32))
        return len(sink)

    fn = CUSTOM_CB(cb)
    keep["fn"] = fn
    return {"lz77_kind": LZ77_CUSTOM, "custom_cb": C.cast(fn, C.c_void_p).value, "custom_level": int(obj.compression_level()),
            "window_size": min(int(obj.window_size()), 32768), "_keep": keep}


class BlockInfo(C.Structure):
    _fields_ = [("start_bit", C.c_uint64), ("end_bit", C.c_uint64), ("btype", C.c_uint32),
                ("bfinal", C.c_uint32), ("out_len", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.lfo_crc32.restype = C.c_uint32
        L.lfo_crc32.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]
        L.lfo_adler32.restype = C.c_uint32
        L.lfo_adler32.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]
        L.lfo_lz77_chunk.restype = C.c_size_t
        L.lfo_lz77_chunk.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p]
        L.lfo_huff_widths.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.lfo_huff_codes.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.lfo_opts_default.argtypes = [C.POINTER(Opts)]
        L.lfo_encoder_new.restype = C.c_void_p
        L.lfo_encoder_new.argtypes = [C.c_int, C.POINTER(Opts)]
        L.lfo_encoder_write.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.lfo_encoder_flush.argtypes = [C.c_void_p]
        L.lfo_encoder_finish.restype = C.POINTER(C.c_uint8)
        L.lfo_encoder_finish.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        L.lfo_encoder_free.argtypes = [C.c_void_p]
        L.lfo_encode_buffer.argtypes = [C.c_int, C.POINTER(Opts), C.c_char_p, C.c_size_t,
                                        C.c_size_t, C.POINTER(Buf)]
        L.lfo_decode.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(Buf),
                                 C.POINTER(C.c_size_t), C.c_char_p]
        L.lfo_buf_free.argtypes = [C.POINTER(Buf)]
        L.lfo_scan_blocks.restype = C.c_long
        L.lfo_scan_blocks.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(BlockInfo), C.c_size_t]
        _lib = L
    return _lib


def crc32(data, crc=0):
    return lib().lfo_crc32(crc, bytes(data), len(data))


def adler32(data, adler=1):
    return lib().lfo_adler32(adler, bytes(data), len(data))


def make_opts(**kw):
    o = Opts()
    lib().lfo_opts_default(C.byref(o))
    for k, v in kw.items():
        if k == "extra":
            o.extra, o.extra_len = v, len(v)
        elif k == "_keep":
            o._keep = v                 # (callback objects of a custom Lz77Encode: alive as long as the options)
        else:
            setattr(o, k, v)
    return o


def lz77_chunk(data, window=32768, max_len=258):
    """→ list of (val, dist) ; dist == 0 is a literal (libflate_lz77 Code)."""
    import numpy as np
    data = bytes(data)
    out = np.zeros(max(len(data), 1), dtype=np.uint32)
    n = lib().lfo_lz77_chunk(data, len(data), window, max_len, out.ctypes.data)
    return out[:n].copy()


def huff_widths(freqs, limit):
    import numpy as np
    f = np.ascontiguousarray(freqs, dtype=np.uint64)
    w = np.zeros(len(f), dtype=np.uint8)
    lib().lfo_huff_widths(f.ctypes.data, len(f), limit, w.ctypes.data)
    return w


def huff_codes(widths):
    import numpy as np
    w = np.ascontiguousarray(widths, dtype=np.uint8)
    b = np.zeros(len(w), dtype=np.uint16)
    lib().lfo_huff_codes(w.ctypes.data, len(w), b.ctypes.data)
    return b


class Encoder:
    """Mirrors {deflate,zlib,gzip}::Encoder: write() = ONE reference write() call."""

    def __init__(self, fmt, **opts):
        self._o = make_opts(**opts)
        self._h = lib().lfo_encoder_new(fmt, C.byref(self._o))

    def write(self, data):
        data = bytes(data)
        lib().lfo_encoder_write(self._h, data, len(data))
        return len(data)

    def flush(self):
        lib().lfo_encoder_flush(self._h)

    def finish(self):
        n = C.c_size_t()
        p = lib().lfo_encoder_finish(self._h, C.byref(n))
        out = C.string_at(p, n.value)
        lib().lfo_encoder_free(self._h)
        self._h = None
        return out


def encode(fmt, data, write_size=0, **opts):
    """write_size == 0: schedule S1 (single write_all); else fixed-size writes (S8K = 8192)."""
    o = make_opts(**opts)
    b = Buf()
    data = bytes(data)
    lib().lfo_encode_buffer(fmt, C.byref(o), data, len(data), write_size, C.byref(b))
    out = C.string_at(b.p, b.n)
    lib().lfo_buf_free(C.byref(b))
    return out


def decode(fmt, data, multi=False):
    """→ (status, output_so_far, consumed, message)"""
    b = Buf()
    data = bytes(data)
    consumed = C.c_size_t()
    err = C.create_string_buffer(160)
    rc = lib().lfo_decode(fmt, 1 if multi else 0, data, len(data), C.byref(b),
                          C.byref(consumed), err)
    out = C.string_at(b.p, b.n) if b.n else b""
    lib().lfo_buf_free(C.byref(b))
    return rc, out, consumed.value, err.value.decode("utf-8", "replace")


def scan_blocks(raw_deflate, max_blocks=1 << 16):
    arr = (BlockInfo * max_blocks)()
    data = bytes(raw_deflate)
    n = lib().lfo_scan_blocks(data, len(data), arr, max_blocks)
    if n < 0:
        raise ValueError("scan_blocks: invalid stream")
    return [(arr[i].start_bit, arr[i].end_bit, arr[i].btype, arr[i].bfinal, arr[i].out_len)
            for i in range(n)]
