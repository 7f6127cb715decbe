"""ctypes binding of the C ABI (include/lfx.h) exported by the in-tree liblfx.so.

There is no Python / CPU fallback: if the HIP library is missing the import fails loudly, and if no
GPU is usable every compute call raises DeviceError.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("LFX_SO") or os.path.join(_HERE, "liblfx.so")     # (LFX_SO: a development build, tools/exp)

DEFLATE, ZLIB, GZIP = 0, 1, 2
OK, E_INVALID_DATA, E_UNEXPECTED_EOF, E_IO, E_OOM, E_DEVICE, E_ARG, E_NOSPACE, E_UNSUPPORTED, E_WOULD_BLOCK = range(10)
LZ77_DEFAULT, LZ77_NOCOMPRESSION = 0, 1
FLUSH_NONE, FLUSH_SYNC = 0, 2
SCHED_SINGLE, SCHED_FIXED, SCHED_LIST = 0, 1, 2
SCHED_FLUSH = (1 << 64) - 1
DEC_MULTI = 1
DEC_NONBLOCKING = 2

# every symbol include/lfx.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "lfx_encode_opts_default", "lfx_ctx_new", "lfx_ctx_free", "lfx_ctx_last_error", "lfx_ctx_set_stream",
    "lfx_device_count", "lfx_encode_bound", "lfx_encode_device", "lfx_encode_batch_device", "lfx_encode_host", "lfx_decode_device",
    "lfx_decode_host", "lfx_decode_batch_device", "lfx_encode_shard_prepare", "lfx_encode_shard_emit", "lfx_encode_shard_prezero", "lfx_decode_shard_device", "lfx_shard_place_device", "lfx_decode_range_scan", "lfx_decode_chain", "lfx_decode_range_emit", "lfx_decode_range_map", "lfx_decode_range_finish",
    "lfx_crc32_combine", "lfx_adler32_combine", "lfx_container_header_len", "lfx_encoder_new",
    "lfx_encoder_write", "lfx_encoder_write_codes", "lfx_encoder_flush", "lfx_encoder_finish", "lfx_encoder_last_error",
    "lfx_encoder_free", "lfx_decoder_new", "lfx_decoder_read", "lfx_decoder_unread",
    "lfx_decoder_consumed", "lfx_decoder_buffered", "lfx_decoder_surplus", "lfx_decoder_header", "lfx_decoder_last_error", "lfx_decoder_free", "lfx_lz77_new",
    "lfx_lz77_encode", "lfx_lz77_flush", "lfx_lz77_window_size", "lfx_lz77_compression_level",
    "lfx_lz77_free", "lfx_ctx_last_timing", "lfx_ctx_enable_timing", "lfx_version",
    "lfx_comm_rccl", "lfx_comm_rccl_free", "lfx_sharded_encode_begin", "lfx_sharded_encode_finish", "lfx_sharded_byte_range",
    "lfx_sharded_decode", "lfx_sharded_layout", "lfx_sharded_gather_tuples", "lfx_sharded_fold", "lfx_sharded_free",
    "lfx_host_alloc", "lfx_host_free", "lfx_ctx_match_fallbacks",
]


class EncodeOpts(C.Structure):
    _fields_ = [
        ("block_size", C.c_uint64), ("dynamic_huffman", C.c_int32), ("no_compression", C.c_int32),
        ("lz77_kind", C.c_int32), ("window_size", C.c_uint32), ("max_length", C.c_uint32),
        ("zlib_flush_mode", C.c_int32), ("mtime", C.c_uint32), ("os", C.c_uint8), ("is_text", C.c_uint8),
        ("hcrc", C.c_uint8), ("lz77_level", C.c_uint8), ("extra", C.c_char_p), ("extra_len", C.c_uint32),
        ("filename", C.c_char_p), ("comment", C.c_char_p),
    ]


class Schedule(C.Structure):
    _fields_ = [("kind", C.c_int32), ("fixed_write", C.c_uint64), ("writes", C.POINTER(C.c_uint64)),
                ("n_writes", C.c_size_t)]


class BlkTuple(C.Structure):
    """lfx_blk_tuple: what the ranks of an N-GPU decode exchange about every candidate block (56 bytes)"""
    _fields_ = [("start_bit", C.c_uint64), ("end_bit", C.c_uint64), ("n_out", C.c_uint64), ("data_bit", C.c_uint64),
                ("n_codes", C.c_uint32), ("nlanes", C.c_uint32), ("btype", C.c_uint8), ("bfinal", C.c_uint8),
                ("status", C.c_uint8), ("_pad", C.c_uint8), ("rank", C.c_uint16), ("_pad2", C.c_uint16),
                ("slot", C.c_uint32), ("_pad3", C.c_uint32)]


class ShardInfo(C.Structure):
    _fields_ = [("total_bits", C.c_uint64), ("n_bytes", C.c_uint64), ("crc32", C.c_uint32),
                ("adler32", C.c_uint32)]


# lfx_comm: the caller's collectives (include/lfx.h); every callback returns 0 on success
COMM_ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)
COMM_P2P = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32)
COMM_WAIT = C.CFUNCTYPE(C.c_int, C.c_void_p)


class Comm(C.Structure):
    _fields_ = [("user", C.c_void_p), ("rank", C.c_uint32), ("world", C.c_uint32), ("allgather", COMM_ALLGATHER),
                ("isend", COMM_P2P), ("irecv", COMM_P2P), ("wait", COMM_WAIT), ("start", COMM_WAIT)]


class ShardedPart(C.Structure):
    _fields_ = [("start_bit", C.c_uint64), ("end_bit", C.c_uint64), ("part_len", C.c_uint64), ("member_len", C.c_uint64), ("total_n", C.c_uint64),
                ("check", C.c_uint32), ("_pad", C.c_uint32)]


class ShardedSlice(C.Structure):
    _fields_ = [("out_len", C.c_uint64), ("out_base", C.c_uint64), ("total_out", C.c_uint64), ("crc32", C.c_uint32),
                ("adler32", C.c_uint32)]


class Header(C.Structure):
    _fields_ = [("format", C.c_int32), ("mtime", C.c_uint32), ("xfl", C.c_uint8), ("os", C.c_uint8),
                ("is_text", C.c_uint8), ("is_verified", C.c_uint8), ("has_extra", C.c_uint8), ("_pad", C.c_uint8 * 3),
                ("extra", C.POINTER(C.c_uint8)), ("extra_len", C.c_uint32), ("filename", C.c_char_p),
                ("comment", C.c_char_p), ("zlib_window_size", C.c_uint32), ("zlib_level", C.c_uint32)]


class Timing(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("phase_ms", C.c_float * 16), ("phase_name", (C.c_char * 24) * 16),
                ("n_phases", C.c_int)]


WRITE_CB = C.CFUNCTYPE(C.c_int64, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)
FLUSH_CB = C.CFUNCTYPE(C.c_int, C.c_void_p)
READ_CB = C.CFUNCTYPE(C.c_int64, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)
SINK_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint32), C.c_size_t)


class LfxError(Exception):
    def __init__(self, status, message=""):
        super().__init__("lfx status %d: %s" % (status, message))
        self.status = status
        self.message = message


class DeviceError(LfxError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            "libflate_amd: native library %s is missing — run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc, gfx950). There is no CPU fallback." % SO_PATH)
    # One HIP runtime per process: PyTorch bundles its own libamdhip64.so.7 / libhsa-runtime64; if
    # /opt/rocm's copy were initialised first, torch (device memory, RCCL) could no longer see the
    # GPU.  Importing torch first makes our DT_NEEDED libamdhip64.so.7 resolve to the loaded one.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(SO_PATH)
    u64, u32, vp, i32 = C.c_uint64, C.c_uint32, C.c_void_p, C.c_int
    L.lfx_encode_opts_default.argtypes = [C.POINTER(EncodeOpts)]
    L.lfx_ctx_new.restype = vp
    L.lfx_ctx_new.argtypes = [i32, C.POINTER(i32)]
    L.lfx_ctx_free.argtypes = [vp]
    L.lfx_ctx_last_error.restype = C.c_char_p
    L.lfx_ctx_last_error.argtypes = [vp]
    L.lfx_ctx_set_stream.argtypes = [vp, vp]
    L.lfx_encode_bound.restype = u64
    L.lfx_encode_bound.argtypes = [u64, C.POINTER(EncodeOpts), C.POINTER(Schedule)]
    L.lfx_encode_device.argtypes = [vp, i32, C.POINTER(EncodeOpts), C.POINTER(Schedule), vp, u64, vp, u64,
                                    C.POINTER(u64)]
    # (host pointers as void*: bytes objects and raw addresses — numpy buffers, lfx_host_alloc blocks — both pass)
    L.lfx_encode_host.argtypes = [vp, i32, C.POINTER(EncodeOpts), C.POINTER(Schedule), vp, u64, vp,
                                  u64, C.POINTER(u64)]
    L.lfx_host_alloc.restype = vp
    L.lfx_host_alloc.argtypes = [C.c_size_t]
    L.lfx_host_free.argtypes = [vp]
    L.lfx_ctx_match_fallbacks.restype = u64
    L.lfx_ctx_match_fallbacks.argtypes = [vp]
    L.lfx_decode_device.argtypes = [vp, i32, u32, vp, u64, vp, u64, C.POINTER(u64), C.POINTER(u64)]
    L.lfx_decode_host.argtypes = [vp, i32, u32, vp, u64, vp, u64, C.POINTER(u64), C.POINTER(u64)]
    L.lfx_decode_batch_device.argtypes = [vp, i32, u32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.lfx_encode_batch_device.argtypes = [vp, i32, C.POINTER(EncodeOpts), C.POINTER(Schedule), u32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.lfx_encode_shard_prepare.argtypes = [vp, i32, C.POINTER(EncodeOpts), C.POINTER(Schedule), vp, u64, i32,
                                           i32, C.POINTER(ShardInfo)]
    L.lfx_encode_shard_emit.argtypes = [vp, u64, u32, u64, vp, u64, C.POINTER(u64)]
    L.lfx_encode_shard_prezero.argtypes = [vp, vp, u64]
    L.lfx_decode_shard_device.argtypes = [vp, vp, u64, u64, u64, i32, vp, u64, C.POINTER(u64)]
    L.lfx_shard_place_device.argtypes = [vp, vp, u64, vp, u64, u64, i32]
    L.lfx_decode_range_scan.argtypes = [vp, vp, u64, u64, u64, u64, u64, u32, C.POINTER(BlkTuple), u32, C.POINTER(u32)]
    L.lfx_decode_chain.argtypes = [C.POINTER(BlkTuple), u32, u64, C.POINTER(u32), u32, C.POINTER(u32), C.POINTER(u64)]
    L.lfx_decode_range_emit.argtypes = [vp, vp, u64, u64, C.POINTER(BlkTuple), C.POINTER(u32), u32, u32, vp, u64,
                                        C.POINTER(u64), C.POINTER(u64), C.POINTER(u32)]
    L.lfx_decode_range_map.argtypes = [vp, vp]
    L.lfx_decode_range_finish.argtypes = [vp, vp, u32, C.POINTER(u32), C.POINTER(u32)]
    L.lfx_crc32_combine.restype = u32
    L.lfx_crc32_combine.argtypes = [u32, u32, u64]
    L.lfx_adler32_combine.restype = u32
    L.lfx_adler32_combine.argtypes = [u32, u32, u64]
    L.lfx_container_header_len.restype = u64
    L.lfx_container_header_len.argtypes = [i32, C.POINTER(EncodeOpts)]
    L.lfx_encoder_new.restype = vp
    L.lfx_encoder_new.argtypes = [vp, i32, C.POINTER(EncodeOpts), WRITE_CB, FLUSH_CB, vp, C.POINTER(i32)]
    L.lfx_encoder_write.restype = C.c_int64
    L.lfx_encoder_write.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.lfx_encoder_write_codes.argtypes = [vp, C.POINTER(u32), C.c_size_t, C.c_char_p, C.c_size_t, i32]
    L.lfx_encoder_flush.argtypes = [vp]
    L.lfx_encoder_finish.argtypes = [vp]
    L.lfx_encoder_last_error.restype = C.c_char_p
    L.lfx_encoder_last_error.argtypes = [vp]
    L.lfx_encoder_free.argtypes = [vp]
    L.lfx_decoder_new.restype = vp
    L.lfx_decoder_new.argtypes = [vp, i32, u32, READ_CB, vp, C.POINTER(i32)]
    L.lfx_decoder_read.restype = C.c_int64
    L.lfx_decoder_read.argtypes = [vp, vp, C.c_size_t]
    L.lfx_decoder_unread.argtypes = [vp, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    L.lfx_decoder_consumed.restype = u64
    L.lfx_decoder_consumed.argtypes = [vp]
    L.lfx_decoder_buffered.restype = u64
    L.lfx_decoder_buffered.argtypes = [vp]
    L.lfx_decoder_surplus.argtypes = [vp, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    L.lfx_decoder_header.argtypes = [vp, C.POINTER(Header)]
    L.lfx_decoder_last_error.restype = C.c_char_p
    L.lfx_decoder_last_error.argtypes = [vp]
    L.lfx_decoder_free.argtypes = [vp]
    L.lfx_lz77_new.restype = vp
    L.lfx_lz77_new.argtypes = [vp, u32, u32, C.POINTER(i32)]
    L.lfx_lz77_encode.argtypes = [vp, C.c_char_p, C.c_size_t, SINK_CB, vp]
    L.lfx_lz77_flush.argtypes = [vp, SINK_CB, vp]
    L.lfx_lz77_window_size.restype = u32
    L.lfx_lz77_window_size.argtypes = [vp]
    L.lfx_lz77_compression_level.argtypes = [vp]
    L.lfx_lz77_free.argtypes = [vp]
    L.lfx_ctx_last_timing.argtypes = [vp, C.POINTER(Timing)]
    L.lfx_ctx_enable_timing.argtypes = [vp, i32]
    L.lfx_version.restype = u32
    L.lfx_comm_rccl.argtypes = [vp, vp, u32, u32, C.POINTER(Comm)]
    L.lfx_comm_rccl_free.argtypes = [C.POINTER(Comm)]
    L.lfx_comm_rccl_free.restype = None
    L.lfx_sharded_encode_begin.argtypes = [vp, C.POINTER(Comm), i32, C.POINTER(EncodeOpts), C.POINTER(Schedule), vp, u64, vp, u64,
                                           vp, u64, vp, u64, C.POINTER(vp), C.POINTER(ShardedPart)]
    L.lfx_sharded_encode_finish.argtypes = [vp, C.POINTER(Comm), vp, C.POINTER(u64)]
    L.lfx_sharded_byte_range.argtypes = [u64, u64, u32, u32, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.lfx_sharded_byte_range.restype = None
    L.lfx_sharded_decode.argtypes = [vp, C.POINTER(Comm), vp, u64, u64, u64, u64, u64, vp, u64, C.POINTER(ShardedSlice)]
    L.lfx_sharded_layout.argtypes = [C.POINTER(Comm), C.POINTER(ShardInfo), u64, i32, C.POINTER(u64), C.POINTER(u32), C.POINTER(u64)]
    L.lfx_sharded_gather_tuples.argtypes = [C.POINTER(Comm), C.POINTER(BlkTuple), u32, i32, C.POINTER(C.POINTER(BlkTuple)),
                                            C.POINTER(u32), C.POINTER(u32)]
    L.lfx_sharded_fold.argtypes = [C.POINTER(Comm), i32, u32, u64, u32, u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32),
                                   C.POINTER(u64), C.POINTER(u32)]
    L.lfx_sharded_free.argtypes = [vp]
    L.lfx_sharded_free.restype = None
    # debug hooks (host execution of host/device-shared code; used by the CPU test-suite only)
    L.lfx_debug_huff_block.argtypes = [vp, u32, vp, vp, vp, C.POINTER(u32), C.POINTER(u64)]
    L.lfx_debug_plan.argtypes = [i32, C.POINTER(EncodeOpts), C.POINTER(Schedule), u64, vp, C.c_size_t,
                                 C.POINTER(C.c_size_t), vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.lfx_debug_plan_incremental.argtypes = [i32, C.POINTER(EncodeOpts), C.POINTER(Schedule), u64, u32, vp, C.c_size_t,
                                             C.POINTER(C.c_size_t), vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.lfx_debug_symbols.argtypes = [u32, u32, vp]
    _lib = L
    return L


def make_opts(**kw):
    """EncodeOptions → lfx_encode_opts; keeps byte strings alive on the returned struct."""
    o = EncodeOpts()
    lib().lfx_encode_opts_default(C.byref(o))
    keep = []
    for k, v in kw.items():
        if k == "extra":
            v = bytes(v)
            keep.append(v)
            o.extra, o.extra_len = v, len(v)
        elif k in ("filename", "comment"):
            v = bytes(v)
            keep.append(v)
            setattr(o, k, v)
        else:
            setattr(o, k, v)
    o._keep = keep
    return o


def make_schedule(write_size=0, writes=None):
    """write_size == 0 and writes is None → one write_all (S1); write_size > 0 → fixed writes (S8K =
    8192); writes = explicit list, None entries mean Write::flush()."""
    s = Schedule()
    if writes is not None:
        arr = (C.c_uint64 * len(writes))(*[SCHED_FLUSH if w is None else int(w) for w in writes])
        s.kind, s.writes, s.n_writes = SCHED_LIST, arr, len(writes)
        s._keep = arr
    elif write_size:
        s.kind, s.fixed_write = SCHED_FIXED, write_size
    else:
        s.kind = SCHED_SINGLE
    return s
