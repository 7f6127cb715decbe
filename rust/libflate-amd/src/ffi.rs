//! Raw bindings of include/lfx.h (bindgen-free: the header is small and stable).  Every function declared here
//! is driven by tests/c/shim_abi.c with the same argument shapes (tests/test_abi.py checks the two lists agree).
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub const LFX_DEFLATE: c_int = 0;
pub const LFX_ZLIB: c_int = 1;
pub const LFX_GZIP: c_int = 2;

// lfx.h: LFX_LEVEL_* (libflate_lz77::CompressionLevel; lfx_encode_opts::lz77_level = 1 + level)
pub const LFX_LEVEL_NONE: c_int = 0;
pub const LFX_LEVEL_FAST: c_int = 1;
pub const LFX_LEVEL_BALANCE: c_int = 2;
pub const LFX_LEVEL_BEST: c_int = 3;

pub const LFX_OK: c_int = 0;
pub const LFX_E_INVALID_DATA: c_int = 1;
pub const LFX_E_UNEXPECTED_EOF: c_int = 2;
pub const LFX_E_IO: c_int = 3;
pub const LFX_E_OOM: c_int = 4;
pub const LFX_E_DEVICE: c_int = 5;
pub const LFX_E_ARG: c_int = 6;
pub const LFX_E_NOSPACE: c_int = 7;
pub const LFX_E_UNSUPPORTED: c_int = 8;
pub const LFX_E_WOULD_BLOCK: c_int = 9;

pub const LFX_LZ77_DEFAULT: i32 = 0;
pub const LFX_LZ77_NOCOMPRESSION: i32 = 1;
pub const LFX_FLUSH_NONE: i32 = 0;
pub const LFX_FLUSH_SYNC: i32 = 2;
pub const LFX_DEC_MULTI: u32 = 1;
pub const LFX_DEC_NONBLOCKING: u32 = 2;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct lfx_encode_opts {
    pub block_size: u64,
    pub dynamic_huffman: i32,
    pub no_compression: i32,
    pub lz77_kind: i32,
    pub window_size: u32,
    pub max_length: u32,
    pub zlib_flush_mode: i32,
    pub mtime: u32,
    pub os: u8,
    pub is_text: u8,
    pub hcrc: u8,
    pub lz77_level: u8,
    pub extra: *const u8,
    pub extra_len: u32,
    pub filename: *const c_char,
    pub comment: *const c_char,
}

#[repr(C)]
pub struct lfx_header {
    pub format: i32,
    pub mtime: u32,
    pub xfl: u8,
    pub os: u8,
    pub is_text: u8,
    pub is_verified: u8,
    pub has_extra: u8,
    pub _pad: [u8; 3],
    pub extra: *const u8,
    pub extra_len: u32,
    pub filename: *const c_char,
    pub comment: *const c_char,
    pub zlib_window_size: u32,
    pub zlib_level: u32,
}

pub enum lfx_ctx {}
pub enum lfx_encoder {}
pub enum lfx_decoder {}
pub enum lfx_lz77 {}

pub type lfx_write_cb = extern "C" fn(user: *mut c_void, p: *const u8, n: usize) -> i64;
pub type lfx_flush_cb = extern "C" fn(user: *mut c_void) -> c_int;
pub type lfx_read_cb = extern "C" fn(user: *mut c_void, p: *mut u8, cap: usize) -> i64;
pub type lfx_sink_cb = extern "C" fn(user: *mut c_void, codes: *const u32, n: usize);

// ---- the N-GPU drivers (include/lfx.h, round 5): the caller's collectives as callbacks (round 6: `start`)
#[repr(C)]
pub struct lfx_comm {
    pub user: *mut c_void,
    pub rank: u32,
    pub world: u32,
    pub allgather: Option<extern "C" fn(user: *mut c_void, send: *const c_void, recv: *mut c_void, bytes: u64) -> c_int>,
    pub isend: Option<extern "C" fn(user: *mut c_void, d_buf: *const c_void, bytes: u64, to_rank: u32) -> c_int>,
    pub irecv: Option<extern "C" fn(user: *mut c_void, d_buf: *mut c_void, bytes: u64, from_rank: u32) -> c_int>,
    pub wait: Option<extern "C" fn(user: *mut c_void) -> c_int>,
    pub start: Option<extern "C" fn(user: *mut c_void) -> c_int>,
}
#[repr(C)]
#[derive(Default, Clone, Copy, Debug)]
pub struct lfx_sharded_part { pub start_bit: u64, pub end_bit: u64, pub part_len: u64, pub member_len: u64, pub total_n: u64, pub check: u32, pub _pad: u32 }
#[repr(C)]
#[derive(Default, Clone, Copy, Debug)]
pub struct lfx_sharded_slice { pub out_len: u64, pub out_base: u64, pub total_out: u64, pub crc32: u32, pub adler32: u32 }
#[repr(C)]
pub struct lfx_sharded_enc { _private: [u8; 0] }
#[repr(C)]
pub struct lfx_schedule { pub kind: c_int, pub fixed_write: u64, pub writes: *const u64, pub n_writes: usize }

extern "C" {
    pub fn lfx_comm_rccl(nccl_comm: *mut c_void, hip_stream: *mut c_void, rank: u32, world: u32, out: *mut lfx_comm) -> c_int;
    pub fn lfx_comm_rccl_free(cm: *mut lfx_comm);
    pub fn lfx_sharded_encode_begin(c: *mut lfx_ctx, cm: *const lfx_comm, format: c_int, o: *const lfx_encode_opts, s: *const lfx_schedule,
                                    d_in: *const c_void, n: u64, d_part: *mut c_void, part_cap: u64, d_member: *mut c_void, member_cap: u64,
                                    d_staging: *mut c_void, staging_cap: u64, state: *mut *mut lfx_sharded_enc, out: *mut lfx_sharded_part) -> c_int;
    pub fn lfx_sharded_encode_finish(c: *mut lfx_ctx, cm: *const lfx_comm, state: *mut lfx_sharded_enc, member_len: *mut u64) -> c_int;
    pub fn lfx_sharded_byte_range(first_byte: u64, member_len: u64, rank: u32, world: u32, lo: *mut u64, hi: *mut u64, hold_hi: *mut u64);
    pub fn lfx_sharded_decode(c: *mut lfx_ctx, cm: *const lfx_comm, d_part: *const c_void, n_part: u64, lo_byte: u64, hi_byte: u64,
                              first_bit: u64, member_len: u64, d_out: *mut c_void, cap: u64, out: *mut lfx_sharded_slice) -> c_int;
    pub fn lfx_container_header_len(format: c_int, o: *const lfx_encode_opts) -> u64;
    pub fn lfx_encode_bound(n: u64, o: *const lfx_encode_opts, s: *const lfx_schedule) -> u64;

    pub fn lfx_version() -> u32;
    pub fn lfx_device_count() -> c_int;
    pub fn lfx_encode_opts_default(o: *mut lfx_encode_opts);
    pub fn lfx_ctx_new(device: c_int, status: *mut c_int) -> *mut lfx_ctx;
    pub fn lfx_ctx_free(c: *mut lfx_ctx);
    pub fn lfx_ctx_last_error(c: *const lfx_ctx) -> *const c_char;
    pub fn lfx_ctx_match_fallbacks(c: *const lfx_ctx) -> u64;
    /// page-locked host memory: buffers handed to lfx_encode_host / lfx_decode_host cross PCIe by DMA without staging
    pub fn lfx_host_alloc(bytes: usize) -> *mut c_void;
    pub fn lfx_host_free(p: *mut c_void);

    pub fn lfx_encoder_new(c: *mut lfx_ctx, format: c_int, o: *const lfx_encode_opts, w: lfx_write_cb,
                           f: Option<lfx_flush_cb>, user: *mut c_void, status: *mut c_int) -> *mut lfx_encoder;
    pub fn lfx_encoder_write(e: *mut lfx_encoder, p: *const u8, n: usize) -> i64;
    pub fn lfx_encoder_write_codes(e: *mut lfx_encoder, codes: *const u32, n_codes: usize, raw: *const u8, n_raw: usize,
                                   end_block: c_int) -> c_int;
    pub fn lfx_encoder_flush(e: *mut lfx_encoder) -> c_int;
    pub fn lfx_encoder_finish(e: *mut lfx_encoder) -> c_int;
    pub fn lfx_encoder_last_error(e: *const lfx_encoder) -> *const c_char;
    pub fn lfx_encoder_free(e: *mut lfx_encoder);

    pub fn lfx_decoder_new(c: *mut lfx_ctx, format: c_int, flags: u32, r: lfx_read_cb, user: *mut c_void,
                           status: *mut c_int) -> *mut lfx_decoder;
    pub fn lfx_decoder_read(d: *mut lfx_decoder, out: *mut u8, cap: usize) -> i64;
    pub fn lfx_decoder_unread(d: *mut lfx_decoder, p: *mut *const u8, n: *mut usize) -> c_int;
    pub fn lfx_decoder_surplus(d: *mut lfx_decoder, p: *mut *const u8, n: *mut usize) -> c_int;
    pub fn lfx_decoder_consumed(d: *const lfx_decoder) -> u64;
    pub fn lfx_decoder_buffered(d: *const lfx_decoder) -> u64;
    pub fn lfx_decoder_header(d: *mut lfx_decoder, h: *mut lfx_header) -> c_int;
    pub fn lfx_decoder_last_error(d: *const lfx_decoder) -> *const c_char;
    pub fn lfx_decoder_free(d: *mut lfx_decoder);

    pub fn lfx_lz77_new(c: *mut lfx_ctx, window_size: u32, max_length: u32, status: *mut c_int) -> *mut lfx_lz77;
    pub fn lfx_lz77_encode(z: *mut lfx_lz77, buf: *const u8, len: usize, sink: lfx_sink_cb, user: *mut c_void) -> c_int;
    pub fn lfx_lz77_flush(z: *mut lfx_lz77, sink: lfx_sink_cb, user: *mut c_void) -> c_int;
    pub fn lfx_lz77_window_size(z: *const lfx_lz77) -> u32;
    pub fn lfx_lz77_compression_level(z: *const lfx_lz77) -> c_int;
    pub fn lfx_lz77_free(z: *mut lfx_lz77);
}
