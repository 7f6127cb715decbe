"""libflate::zlib — Encoder, Decoder, EncodeOptions, FlushMode (reference src/zlib.rs)."""
from . import _ffi
from . import deflate as _deflate
from ._stream import StreamError, _DecoderBase, _EncoderBase  # noqa: F401


class FlushMode:  # zlib.rs:184-195
    NONE = _ffi.FLUSH_NONE
    SYNC = _ffi.FLUSH_SYNC


class EncodeOptions(_deflate.EncodeOptions):
    """zlib::EncodeOptions (zlib.rs:414-518)."""

    def flush_mode(self, mode):  # zlib.rs:504-507
        self._kw["zlib_flush_mode"] = mode
        return self


class Encoder(_EncoderBase):
    """zlib::Encoder (zlib.rs:522-681)."""
    FORMAT = _ffi.ZLIB

    @classmethod
    def new(cls, inner, context=None):
        return cls(inner, None, context)

    @classmethod
    def with_options(cls, inner, options, context=None):
        return cls(inner, options, context)


class Decoder(_DecoderBase):
    """zlib::Decoder (zlib.rs:284-410)."""
    FORMAT = _ffi.ZLIB

    @classmethod
    def new(cls, inner, context=None):
        return cls(inner, context)
