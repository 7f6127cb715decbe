"""Device context (lfx_ctx): one per GPU, owns a HIP stream and cached HBM scratch."""
import ctypes as C

from . import _ffi


class Context:
    def __init__(self, device=0):
        st = C.c_int(0)
        self._h = _ffi.lib().lfx_ctx_new(device, C.byref(st))
        if not self._h:
            raise _ffi.DeviceError(st.value, "no usable MI355X device %d (there is no CPU fallback)" % device)
        self.device = device
        self._users = 0          # native handles (encoders, decoders, LZ77 encoders) made from this context and not yet freed
        self._closing = False

    # A native handle uses its context until it is freed (it takes the context's mutex in lfx_*_free).  A Python reference from
    # the handle's wrapper to this object is not enough: the garbage collector runs the finalizers of an unreachable group in
    # an undefined order (PEP 442) — at interpreter exit Context.__del__ ran before a decoder's, whose free then locked a mutex
    # in freed memory ("std::system_error: Invalid argument" under MALLOC_PERTURB_, a corrupted heap without).  So the wrappers
    # count themselves in and out, and the native context is freed by whoever comes last.
    def _retain(self):
        self._users += 1

    def _release(self):
        self._users -= 1
        if self._users <= 0 and self._closing:
            self._free_now()

    def _free_now(self):
        if self._h:
            _ffi.lib().lfx_ctx_free(self._h)
            self._h = None

    def close(self):
        self._closing = True
        if self._users <= 0:
            self._free_now()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def last_error(self):
        return (_ffi.lib().lfx_ctx_last_error(self._h) or b"").decode("utf-8", "replace")

    def set_stream(self, hip_stream_ptr):
        _ffi.lib().lfx_ctx_set_stream(self._h, hip_stream_ptr)

    def enable_timing(self, on=True):
        """True / 1: an event behind every phase; 2..5: only the two events around one kernel's phase (2: lz77_walk, 3: blk_scan,
        4: lz77_cand, 5: lz77_copy); 6: every phase, the encode's match and parse phases split by kernel; False / 0: none."""
        _ffi.lib().lfx_ctx_enable_timing(self._h, int(on) if on is not True and on is not False and 2 <= int(on) <= 6 else (1 if on else 0))

    def last_timing(self):
        t = _ffi.Timing()
        if _ffi.lib().lfx_ctx_last_timing(self._h, C.byref(t)) != 0:
            return None
        return {"total_ms": t.total_ms,
                "phases": [(bytes(t.phase_name[i]).split(b"\0")[0].decode(), t.phase_ms[i])
                           for i in range(t.n_phases)]}

    def match_fallbacks(self):
        """encode passes that ran on the fallback match kernel (lfx_ctx_match_fallbacks): 0 unless the default kernel's
        hardware assumption was seen violated on this part"""
        return int(_ffi.lib().lfx_ctx_match_fallbacks(self._h))

    # ---- one-shot calls on raw HOST pointers (ints: numpy buffers, lfx_host_alloc blocks) — no Python-side copies --------
    def encode_host_ptr(self, fmt, in_ptr, n, out_ptr, cap, opts=None, schedule=None):
        out_len = C.c_uint64(0)
        rc = _ffi.lib().lfx_encode_host(self._h, fmt, C.byref(opts) if opts is not None else None,
                                        C.byref(schedule) if schedule is not None else None, in_ptr, n, out_ptr, cap, C.byref(out_len))
        if rc:
            raise (_ffi.DeviceError if rc == _ffi.E_DEVICE else _ffi.LfxError)(rc, self.last_error())
        return out_len.value

    def decode_host_ptr(self, fmt, in_ptr, n, out_ptr, cap, flags=0):
        """→ (status, out_len, consumed, message)"""
        out_len, consumed = C.c_uint64(0), C.c_uint64(0)
        rc = _ffi.lib().lfx_decode_host(self._h, fmt, flags, in_ptr, n, out_ptr, cap, C.byref(out_len), C.byref(consumed))
        if rc in (_ffi.E_DEVICE, _ffi.E_OOM, _ffi.E_ARG):
            raise (_ffi.DeviceError if rc == _ffi.E_DEVICE else _ffi.LfxError)(rc, self.last_error())
        return rc, out_len.value, consumed.value, self.last_error() if rc else ""

    # ---- one-shot calls on raw device pointers (ints) -------------------------------------------
    def encode_device(self, fmt, d_in, n, d_out, cap, opts=None, schedule=None):
        out_len = C.c_uint64(0)
        rc = _ffi.lib().lfx_encode_device(self._h, fmt, C.byref(opts) if opts is not None else None,
                                          C.byref(schedule) if schedule is not None else None,
                                          d_in, n, d_out, cap, C.byref(out_len))
        if rc:
            raise (_ffi.DeviceError if rc == _ffi.E_DEVICE else _ffi.LfxError)(rc, self.last_error())
        return out_len.value

    def decode_device(self, fmt, d_in, n, d_out, cap, flags=0):
        """→ (status, out_len, consumed, message)"""
        out_len, consumed = C.c_uint64(0), C.c_uint64(0)
        rc = _ffi.lib().lfx_decode_device(self._h, fmt, flags, d_in, n, d_out, cap, C.byref(out_len),
                                          C.byref(consumed))
        if rc in (_ffi.E_DEVICE, _ffi.E_OOM, _ffi.E_ARG):
            raise (_ffi.DeviceError if rc == _ffi.E_DEVICE else _ffi.LfxError)(rc, self.last_error())
        return rc, out_len.value, consumed.value, self.last_error() if rc else ""

    def encode_host(self, fmt, data, opts=None, schedule=None):
        data = bytes(data)
        bound = _ffi.lib().lfx_encode_bound(len(data), C.byref(opts) if opts is not None else None,
                                            C.byref(schedule) if schedule is not None else None)
        if bound == 0:
            raise _ffi.LfxError(_ffi.E_ARG, "option outside the reference's domain")
        out = C.create_string_buffer(bound)
        out_len = C.c_uint64(0)
        rc = _ffi.lib().lfx_encode_host(self._h, fmt, C.byref(opts) if opts is not None else None,
                                        C.byref(schedule) if schedule is not None else None,
                                        data, len(data), out, bound, C.byref(out_len))
        if rc:
            raise (_ffi.DeviceError if rc == _ffi.E_DEVICE else _ffi.LfxError)(rc, self.last_error())
        return out.raw[:out_len.value]

    def decode_host(self, fmt, data, cap=None, flags=0):
        """→ (status, output_so_far, consumed, message)"""
        data = bytes(data)
        cap = cap if cap is not None else max(1 << 16, len(data) * 16)
        while True:
            out = C.create_string_buffer(cap)
            out_len, consumed = C.c_uint64(0), C.c_uint64(0)
            rc = _ffi.lib().lfx_decode_host(self._h, fmt, flags, data, len(data), out, cap,
                                            C.byref(out_len), C.byref(consumed))
            if rc in (_ffi.E_DEVICE, _ffi.E_OOM, _ffi.E_ARG):
                raise (_ffi.DeviceError if rc == _ffi.E_DEVICE else _ffi.LfxError)(rc, self.last_error())
            if rc == _ffi.E_NOSPACE and cap < len(data) * 1040 + (1 << 20):
                cap *= 8
                continue
            return rc, out.raw[:out_len.value], consumed.value, self.last_error() if rc else ""


_default = {}


def default_context(device=0):
    if device not in _default:
        _default[device] = Context(device)
    return _default[device]
