"""Shared implementation of the io::Write / io::Read shaped stream objects over lfx_encoder /
lfx_decoder (include/lfx.h)."""
import ctypes as C

from . import _ffi
from .context import default_context

_KIND = {_ffi.E_INVALID_DATA: "InvalidData", _ffi.E_UNEXPECTED_EOF: "UnexpectedEof"}


class StreamError(IOError):
    """io::Error: .kind is 'InvalidData' / 'UnexpectedEof' / 'Other' (src/lib.rs:10-29)."""

    def __init__(self, status, message):
        super().__init__(message)
        self.status = status
        self.kind = _KIND.get(status, "Other")
        self.message = message


class _EncoderBase:
    FORMAT = _ffi.DEFLATE

    def __init__(self, inner, options=None, context=None):
        """`inner` is any object with write(bytes) (and optionally flush()) — the reference's `W`."""
        self._ctx = context or default_context()
        self._inner = inner
        self._opts = options._to_c() if options is not None else _ffi.make_opts()
        # a caller-supplied Lz77Encode (EncodeOptions::with_lz77(E), encode.rs:59-65) runs here; its codes go to the GPU
        self._lz77 = getattr(options, "_foreign", None) if options is not None else None
        if self._lz77 is not None and self._opts.no_compression:
            self._lz77 = None                    # RawBuf never calls the Lz77Encode (encode.rs:348-383)
        self._original_size = 0
        self._wcb = _ffi.WRITE_CB(self._on_write)
        self._fcb = _ffi.FLUSH_CB(self._on_flush)
        st = C.c_int(0)
        self._h = _ffi.lib().lfx_encoder_new(self._ctx.handle, self.FORMAT, C.byref(self._opts), self._wcb,
                                             self._fcb, None, C.byref(st))
        if not self._h:
            raise StreamError(st.value, "encoder construction failed (status %d)" % st.value)
        self._ctx._retain()

    def _on_write(self, _user, p, n):
        try:
            self._inner.write(C.string_at(p, n))
            return n
        except Exception:
            return -5

    def _on_flush(self, _user):
        try:
            f = getattr(self._inner, "flush", None)
            if f:
                f()
            return 0
        except Exception:
            return 1

    def write(self, buf):
        """io::Write::write — ONE reference write() call; always consumes everything."""
        buf = bytes(buf)
        if self._lz77 is not None:
            sink = []
            self._lz77.encode(buf, sink)                         # CompressBuf::append encode.rs:405-408
            self._codes(sink, buf, 0)
            self._original_size += len(buf)
            while self._original_size >= self._opts.block_size:  # Block::write encode.rs:282
                self._close_block(1)
            return len(buf)
        r = _ffi.lib().lfx_encoder_write(self._h, buf, len(buf))
        if r < 0:
            raise StreamError(-r, self._err())
        return r

    def _codes(self, sink, raw, end_block):
        from .lz77 import Code
        words = (C.c_uint32 * max(len(sink), 1))(*[Code.to_word(c) for c in sink])
        rc = _ffi.lib().lfx_encoder_write_codes(self._h, words, len(sink), raw, len(raw), end_block)
        if rc:
            raise StreamError(rc, self._err())

    def _close_block(self, end_block):
        sink = []
        self._lz77.flush(sink)                                   # CompressBuf::flush encode.rs:416
        self._codes(sink, b"", end_block)
        self._original_size = 0

    write_all = write

    def flush(self):
        if self._lz77 is not None:
            self._close_block(1)
        rc = _ffi.lib().lfx_encoder_flush(self._h)
        if rc:
            raise StreamError(rc, self._err())

    def finish(self):
        """Encoder::finish → the inner writer (Finish<W, io::Error>::into_result)."""
        if self._lz77 is not None:
            self._close_block(2)
        rc = _ffi.lib().lfx_encoder_finish(self._h)
        msg = self._err()
        self._free()
        if rc:
            raise StreamError(rc, msg)
        return self._inner

    def as_inner_ref(self):
        return self._inner

    def into_inner(self):
        self._free()
        return self._inner

    def _err(self):
        return (_ffi.lib().lfx_encoder_last_error(self._h) or b"").decode("utf-8", "replace")

    def _free(self):
        """frees the native handle, then lets go of the context (Context._retain / _release: the context outlives its handles
        whatever the order of the finalizers)"""
        if getattr(self, "_h", None):
            _ffi.lib().lfx_encoder_free(self._h)
            self._h = None
            self._ctx._release()

    def __del__(self):
        self._free()


class _DecoderBase:
    FORMAT = _ffi.DEFLATE
    FLAGS = 0

    def __init__(self, inner, context=None):
        """`inner`: bytes-like or an object with read(n) — the reference's `R`."""
        self._ctx = context or default_context()
        if isinstance(inner, (bytes, bytearray, memoryview)):
            import io
            inner = io.BytesIO(bytes(inner))
        self._inner = inner
        self._rcb = _ffi.READ_CB(self._on_read)
        st = C.c_int(0)
        self._h = _ffi.lib().lfx_decoder_new(self._ctx.handle, self.FORMAT, self.FLAGS, self._rcb, None,
                                             C.byref(st))
        if not self._h:
            raise StreamError(st.value, self._ctx.last_error())
        self._ctx._retain()

    def _on_read(self, _user, p, cap):
        try:
            b = self._inner.read(cap)
            if b is None:                      # a non-blocking raw stream with nothing to hand over
                return -_ffi.E_WOULD_BLOCK
            C.memmove(p, b, len(b))
            return len(b)
        except BlockingIOError:
            return -_ffi.E_WOULD_BLOCK
        except Exception:
            return -5

    def read(self, n=-1):
        """io::Read::read (n >= 0) or read_to_end (n < 0).  Raises StreamError on failure."""
        if n == 0:
            return b""
        chunks = []
        want = n
        while True:
            cap = 1 << 20 if want < 0 else want
            buf = C.create_string_buffer(cap)
            r = _ffi.lib().lfx_decoder_read(self._h, buf, cap)
            if r == -_ffi.E_WOULD_BLOCK:       # io::ErrorKind::WouldBlock: call again later (non-blocking decoders)
                if chunks:
                    break
                raise BlockingIOError("WouldBlock")
            if r < 0:
                err = StreamError(-r, self._err())
                err.partial = b"".join(chunks)
                raise err
            if r == 0:
                break
            chunks.append(buf.raw[:r])
            if want >= 0:
                break
        return b"".join(chunks)

    def read_to_end(self):
        return self.read(-1)

    def unread_decoded_data(self):
        p = C.POINTER(C.c_uint8)()
        n = C.c_size_t(0)
        _ffi.lib().lfx_decoder_unread(self._h, C.byref(p), C.byref(n))
        return C.string_at(p, n.value) if n.value else b""

    def header(self):
        """gzip::Header / zlib::Header of the current member (gzip.rs:292-341, zlib.rs:197-220) as a dict."""
        h = _ffi.Header()
        rc = _ffi.lib().lfx_decoder_header(self._h, C.byref(h))
        if rc == _ffi.E_WOULD_BLOCK:
            raise BlockingIOError("WouldBlock")
        if rc:
            raise StreamError(rc, self._err())
        if self.FORMAT == _ffi.ZLIB:
            return {"window_size": h.zlib_window_size, "compression_level": h.zlib_level}
        return {"modification_time": h.mtime, "xfl": h.xfl, "os": h.os, "is_text": bool(h.is_text),
                "is_verified": bool(h.is_verified),
                "extra_field": C.string_at(h.extra, h.extra_len) if h.has_extra else None,
                "filename": h.filename, "comment": h.comment}

    def surplus(self):
        """bytes already pulled from the inner reader that lie behind the last finished member — what a caller hands
        to whatever reads next after into_inner() (gzip.rs:1216-1226)."""
        p = C.POINTER(C.c_uint8)()
        n = C.c_size_t(0)
        _ffi.lib().lfx_decoder_surplus(self._h, C.byref(p), C.byref(n))
        return C.string_at(p, n.value) if n.value else b""

    def consumed(self):
        """bytes of the inner reader that belong to the decoded stream (into_inner position)."""
        return _ffi.lib().lfx_decoder_consumed(self._h)

    def _err(self):
        return (_ffi.lib().lfx_decoder_last_error(self._h) or b"").decode("utf-8", "replace")

    def __del__(self):
        if getattr(self, "_h", None):
            _ffi.lib().lfx_decoder_free(self._h)
            self._h = None
            self._ctx._release()
