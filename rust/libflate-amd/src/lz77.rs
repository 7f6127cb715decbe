//! `impl libflate_lz77::Lz77Encode for GpuLz77Encoder` — the plug-in point of the ORIGINAL crate
//! (`EncodeOptions::with_lz77`, src/deflate/encode.rs:59-65; trait: libflate_lz77/src/lib.rs:83-107).
//!
//! ```ignore
//! let options = libflate::deflate::EncodeOptions::with_lz77(libflate_amd::lz77::GpuLz77Encoder::new()?);
//! let mut encoder = libflate::deflate::Encoder::with_options(Vec::new(), options);
//! ```
//! Only the match search + greedy parse run on the GPU this way (one chunk per `flush`, i.e. per >= window*8
//! buffered bytes, default.rs:60-68); Huffman coding stays in the reference.  The full-path encoders of this crate
//! (`gzip::Encoder` …) are the fast route; this one is the minimal drop-in.
use crate::{default_context, ffi, io_error, Context};
use libflate_lz77::{Code, CompressionLevel, Lz77Encode, Sink};
use std::io;
use std::os::raw::{c_int, c_void};
use std::sync::Arc;

pub struct GpuLz77Encoder {
    h: *mut ffi::lfx_lz77,
    _ctx: Arc<Context>,
}
unsafe impl Send for GpuLz77Encoder {}

impl GpuLz77Encoder {
    /// `DefaultLz77Encoder::new` (default.rs:32-36): window 32768, max length 258
    pub fn new() -> io::Result<Self> { Self::with_window_size_and_max_length(32768, 258) }
    /// `DefaultLz77EncoderBuilder` (default.rs:202-249); values are clamped like the reference's
    pub fn with_window_size_and_max_length(window_size: u16, max_length: u16) -> io::Result<Self> {
        let ctx = default_context()?;
        let mut st: c_int = 0;
        let w = if window_size == 0 { 32768 } else { window_size as u32 };
        let h = unsafe { ffi::lfx_lz77_new(ctx.0, w, max_length as u32, &mut st) };
        if h.is_null() {
            return Err(io_error(st, "lfx_lz77_new failed".to_string()));
        }
        Ok(GpuLz77Encoder { h, _ctx: ctx })
    }
}
impl Drop for GpuLz77Encoder {
    fn drop(&mut self) { unsafe { ffi::lfx_lz77_free(self.h) } }
}

// code word = (val << 16) | dist ; dist == 0 → Literal(val), else Pointer { length: val, backward_distance: dist }
extern "C" fn sink_tramp<S: Sink>(user: *mut c_void, codes: *const u32, n: usize) {
    let sink = unsafe { &mut *(user as *mut S) };
    for &w in unsafe { std::slice::from_raw_parts(codes, n) } {
        let (val, dist) = ((w >> 16) as u16, (w & 0xFFFF) as u16);
        if dist == 0 {
            sink.consume(Code::Literal(val as u8));
        } else {
            sink.consume(Code::Pointer { length: val, backward_distance: dist });
        }
    }
}

impl Lz77Encode for GpuLz77Encoder {
    fn encode<S: Sink>(&mut self, buf: &[u8], mut sink: S) {
        // infallible like the trait: a device failure here is a programming / environment error
        let st = unsafe { ffi::lfx_lz77_encode(self.h, buf.as_ptr(), buf.len(), sink_tramp::<S>, &mut sink as *mut S as *mut c_void) };
        assert_eq!(st, ffi::LFX_OK, "lfx_lz77_encode failed");
    }
    fn flush<S: Sink>(&mut self, mut sink: S) {
        let st = unsafe { ffi::lfx_lz77_flush(self.h, sink_tramp::<S>, &mut sink as *mut S as *mut c_void) };
        assert_eq!(st, ffi::LFX_OK, "lfx_lz77_flush failed");
    }
    fn compression_level(&self) -> CompressionLevel {
        match unsafe { ffi::lfx_lz77_compression_level(self.h) } {
            0 => CompressionLevel::None,
            1 => CompressionLevel::Fast,
            3 => CompressionLevel::Best,
            _ => CompressionLevel::Balance,
        }
    }
    fn window_size(&self) -> u16 {
        // (the trait returns u16: 32768 does not fit and the reference returns MAX_WINDOW_SIZE = 0x8000 as u16)
        unsafe { ffi::lfx_lz77_window_size(self.h) as u16 }
    }
}
