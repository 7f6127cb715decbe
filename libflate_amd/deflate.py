"""libflate::deflate — Encoder, Decoder, EncodeOptions, DEFAULT_BLOCK_SIZE
(reference src/deflate/mod.rs:22-25, src/deflate/encode.rs, src/deflate/decode.rs)."""
from . import _ffi
from ._stream import StreamError, _DecoderBase, _EncoderBase  # noqa: F401

DEFAULT_BLOCK_SIZE = 1024 * 1024  # encode.rs:11


class EncodeOptions:
    """deflate::EncodeOptions (encode.rs:17-128): builder methods return self."""

    def __init__(self, lz77=None):
        self._kw = {}
        self._foreign = None
        if lz77 is not None:
            self.with_lz77(lz77)

    @classmethod
    def new(cls):
        return cls()

    def with_lz77(self, lz77):  # encode.rs:59-65
        """`lz77`: one of this package's encoders (DefaultLz77Encoder, NoCompressionLz77Encoder: the whole path runs on the
        GPU) or ANY object with the Lz77Encode methods encode(buf, sink) / flush(sink) / compression_level() /
        window_size() (lib.rs:83-107): it then runs on the caller's side and the GPU Huffman-codes what it emits
        (lfx_encoder_write_codes)."""
        if hasattr(lz77, "_opts"):
            self._kw.update(lz77._opts())
            self._foreign = None
        else:
            self._foreign = lz77
            self._kw.update({"lz77_kind": _ffi.LZ77_DEFAULT, "window_size": min(int(lz77.window_size()), 32768),
                             "lz77_level": 1 + int(lz77.compression_level())})
        return self

    def no_compression(self):  # encode.rs:77-80
        self._kw["no_compression"] = 1
        return self

    def block_size(self, size):  # encode.rs:93-96
        self._kw["block_size"] = size
        return self

    def fixed_huffman_codes(self):  # encode.rs:107-110
        self._kw["dynamic_huffman"] = 0
        return self

    def _to_c(self):
        return _ffi.make_opts(**self._kw)


class Encoder(_EncoderBase):
    """deflate::Encoder (encode.rs:130-249)."""
    FORMAT = _ffi.DEFLATE

    @classmethod
    def new(cls, inner, context=None):
        return cls(inner, None, context)

    @classmethod
    def with_options(cls, inner, options, context=None):
        return cls(inner, options, context)


class Decoder(_DecoderBase):
    """deflate::Decoder (decode.rs:8-164)."""
    FORMAT = _ffi.DEFLATE

    @classmethod
    def new(cls, inner, context=None):
        return cls(inner, context)
