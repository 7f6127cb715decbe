"""libflate::lz77 / libflate_lz77 — Code, Lz77Encode implementations (reference
libflate_lz77/src/lib.rs, libflate_lz77/src/default.rs)."""
import ctypes as C

from . import _ffi
from .context import default_context

MAX_LENGTH = 258        # lib.rs:18
MAX_DISTANCE = 32768    # lib.rs:21
MAX_WINDOW_SIZE = MAX_DISTANCE


class Code:
    """lz77::Code (lib.rs:27-42)."""

    @staticmethod
    def Literal(b):
        return ("Literal", b)

    @staticmethod
    def Pointer(length, backward_distance):
        return ("Pointer", length, backward_distance)

    @staticmethod
    def to_word(code):
        """the ABI's code word: (val << 16) | dist, dist == 0 for a literal (include/lfx.h, lfx_sink_cb)"""
        return (code[1] << 16) if code[0] == "Literal" else (code[1] << 16) | code[2]

    @staticmethod
    def from_word(w):
        w = int(w)
        return Code.Literal(w >> 16) if (w & 0xFFFF) == 0 else Code.Pointer(w >> 16, w & 0xFFFF)


class CompressionLevel:  # lib.rs:44-58
    NONE, FAST, BALANCE, BEST = 0, 1, 2, 3


class DefaultLz77Encoder:
    """Lz77Encode implementation used by default (default.rs:14-109), match finding on the GPU."""

    def __init__(self, window_size=MAX_WINDOW_SIZE, max_length=MAX_LENGTH, context=None):
        self._ctx = context or default_context()
        self._window = min(window_size, MAX_WINDOW_SIZE)
        self._max_length = min(max_length, MAX_LENGTH)
        st = C.c_int(0)
        self._h = _ffi.lib().lfx_lz77_new(self._ctx.handle, self._window, self._max_length, C.byref(st))
        if not self._h:
            raise _ffi.LfxError(st.value, "lfx_lz77_new failed")
        self._ctx._retain()      # (the context outlives its handles whatever the order of the finalizers: Context._retain)

    @classmethod
    def new(cls):
        return cls()

    @classmethod
    def with_window_size(cls, size):
        return cls(window_size=size)

    def _run(self, fn, *args):
        got = []

        def sink(_u, p, n):
            got.extend(p[i] for i in range(n))

        cb = _ffi.SINK_CB(sink)
        rc = fn(self._h, *args, cb, None)
        if rc:
            raise _ffi.LfxError(rc, self._ctx.last_error())
        return got

    def encode(self, buf, sink):
        """Lz77Encode::encode: `sink` is a list (Vec<Code>) that receives Code tuples."""
        buf = bytes(buf)
        sink.extend(Code.from_word(w) for w in self._run(_ffi.lib().lfx_lz77_encode, buf, len(buf)))

    def flush(self, sink):
        sink.extend(Code.from_word(w) for w in self._run(_ffi.lib().lfx_lz77_flush))

    def window_size(self):
        return _ffi.lib().lfx_lz77_window_size(self._h)

    def compression_level(self):
        return _ffi.lib().lfx_lz77_compression_level(self._h)

    def _opts(self):
        return {"lz77_kind": _ffi.LZ77_DEFAULT, "window_size": self._window, "max_length": self._max_length}

    def __del__(self):
        if getattr(self, "_h", None):
            _ffi.lib().lfx_lz77_free(self._h)
            self._h = None
            self._ctx._release()


class DefaultLz77EncoderBuilder:  # default.rs:202-249
    def __init__(self):
        self._w, self._m = MAX_WINDOW_SIZE, MAX_LENGTH

    @classmethod
    def new(cls):
        return cls()

    def window_size(self, w):
        self._w = min(w, MAX_WINDOW_SIZE)
        return self

    def max_length(self, m):
        self._m = min(m, MAX_LENGTH)
        return self

    def build(self):
        return DefaultLz77Encoder(self._w, self._m)


class NoCompressionLz77Encoder:
    """lib.rs:111-145: every byte a literal (still Huffman coded by the block encoder)."""

    @classmethod
    def new(cls):
        return cls()

    def compression_level(self):
        return CompressionLevel.NONE

    def window_size(self):
        return MAX_WINDOW_SIZE

    def _opts(self):
        return {"lz77_kind": _ffi.LZ77_NOCOMPRESSION}
