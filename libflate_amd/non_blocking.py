"""libflate::non_blocking::{deflate,zlib,gzip}::Decoder (reference src/non_blocking/): decoders over readers that
may answer io::ErrorKind::WouldBlock.  `inner.read(n)` raising BlockingIOError (or returning None) maps to
WouldBlock; Decoder.read() / header() then raise BlockingIOError and can simply be called again — the state is kept
(src/non_blocking/deflate/decode.rs:66-147 rolls its bit reader back to the last transaction; here the bytes seen
so far stay buffered and the GPU decode is attempted whenever the reader has nothing more for the moment)."""
from . import _ffi
from ._stream import StreamError, _DecoderBase  # noqa: F401


class _NB(_DecoderBase):
    FLAGS = _ffi.DEC_NONBLOCKING

    @classmethod
    def new(cls, inner, context=None):
        return cls(inner, context)


class deflate:  # noqa: N801  (module-like namespace: non_blocking::deflate::Decoder)
    class Decoder(_NB):
        FORMAT = _ffi.DEFLATE


class zlib:  # noqa: N801
    class Decoder(_NB):
        FORMAT = _ffi.ZLIB


class gzip:  # noqa: N801
    class Decoder(_NB):
        FORMAT = _ffi.GZIP
