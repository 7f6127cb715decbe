"""libflate::gzip — Encoder, Decoder, MultiDecoder, EncodeOptions, HeaderBuilder (reference src/gzip.rs)."""
import struct

from . import _ffi
from . import deflate as _deflate
from ._stream import StreamError, _DecoderBase, _EncoderBase  # noqa: F401


class ExtraSubField:  # gzip.rs:507-541
    def __init__(self, id, data):
        self.id = bytes(id)
        self.data = bytes(data)

    def _bytes(self):
        return self.id + struct.pack("<H", len(self.data)) + self.data


class HeaderBuilder:
    """gzip::HeaderBuilder (gzip.rs:126-288).  modification_time defaults to 0 here (the reference
    uses `now`, gzip.rs:148-151) — set it explicitly for reproducible bytes."""

    def __init__(self):
        self._kw = {"mtime": 0}

    @classmethod
    def new(cls):
        return cls()

    def modification_time(self, t):
        self._kw["mtime"] = t
        return self

    def os(self, os_code):
        self._kw["os"] = os_code
        return self

    def text(self):
        self._kw["is_text"] = 1
        return self

    def verify(self):
        self._kw["hcrc"] = 1
        return self

    def extra_field(self, subfields):
        self._kw["extra"] = b"".join(f._bytes() for f in subfields)
        return self

    def filename(self, name):
        self._kw["filename"] = bytes(name)
        return self

    def comment(self, text):
        self._kw["comment"] = bytes(text)
        return self

    def finish(self):
        return dict(self._kw)


class EncodeOptions(_deflate.EncodeOptions):
    """gzip::EncodeOptions (gzip.rs:639-751)."""

    def __init__(self, lz77=None):
        super().__init__(lz77)
        self._kw.setdefault("mtime", 0)

    def no_compression(self):  # gzip.rs:700-705
        """stored blocks; the header's compression level goes back to Unknown at this point (gzip.rs:703) — a header()
        call AFTER it brings its own level along again"""
        super().no_compression()
        self._kw["lz77_level"] = 1 + 2
        return self

    def header(self, header):  # gzip.rs:717-720
        """`header`: what HeaderBuilder.finish() returns, or the dict a Decoder's header() returns (gzip.rs:959).  The header
        REPLACES the one the options held, its compression level included: the XFL byte written is the header's
        own — Unknown for a builder's header (gzip.rs:157), Fastest / Slowest kept for one cloned from a decoder."""
        h = dict(header)
        if "modification_time" in h:        # a decoder's header: the reference's field names
            h = {k: v for k, v in {"mtime": h["modification_time"], "os": h.get("os", 3), "is_text": int(bool(h.get("is_text"))),
                                   "hcrc": int(bool(h.get("is_verified"))), "extra": h.get("extra_field"),
                                   "filename": h.get("filename"), "comment": h.get("comment"), "xfl": h.get("xfl", 0)}.items()
                 if v is not None}
        xfl = h.pop("xfl", 0)
        self._kw.update(h)
        # (lz77_level = 1 + libflate_lz77::CompressionLevel: Fast → XFL 4, Best → XFL 2, Balance → XFL 0, gzip.rs:84-92)
        self._kw["lz77_level"] = {4: 1 + 1, 2: 1 + 3}.get(xfl, 1 + 2)
        return self


class Encoder(_EncoderBase):
    """gzip::Encoder (gzip.rs:754-908)."""
    FORMAT = _ffi.GZIP

    def __init__(self, inner, options=None, context=None):
        super().__init__(inner, options if options is not None else EncodeOptions(), context)

    @classmethod
    def new(cls, inner, context=None):
        return cls(inner, None, context)

    @classmethod
    def with_options(cls, inner, options, context=None):
        return cls(inner, options, context)


class Decoder(_DecoderBase):
    """gzip::Decoder (gzip.rs:912-1047): one member; trailing bytes are left unread."""
    FORMAT = _ffi.GZIP

    @classmethod
    def new(cls, inner, context=None):
        return cls(inner, context)


class MultiDecoder(_DecoderBase):
    """gzip::MultiDecoder (gzip.rs:1052-1167): concatenated members."""
    FORMAT = _ffi.GZIP
    FLAGS = _ffi.DEC_MULTI

    @classmethod
    def new(cls, inner, context=None):
        return cls(inner, context)
