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
pub use libflate_lz77::{Code, CompressionLevel, Lz77Encode, Sink, MAX_DISTANCE, MAX_LENGTH, MAX_WINDOW_SIZE};
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

// ------------------------------------------------------------------------------------------------------------------
// The `E` of `EncodeOptions<E>` / `Encoder<W, E>` (src/deflate/encode.rs:17,132; src/gzip.rs:639,754; src/zlib.rs:414,522).
//
// The reference runs ANY `E: Lz77Encode` on the CPU and Huffman-codes what it emits.  Here the whole path — match search,
// parse, Huffman, bit packing — is one GPU pipeline, so `E` has to be an LZ77 stage that pipeline implements: the trait
// `GpuLz77` below (sealed) is the bound, implemented by this module's `DefaultLz77Encoder` and
// `NoCompressionLz77Encoder` — the two implementations the reference ships (default.rs:14-109, lib.rs:111-145), under the
// reference's names, so that `EncodeOptions::with_lz77(DefaultLz77EncoderBuilder::new().window_size(1024).build())`
// compiles unchanged.  A user-defined `E: Lz77Encode` is a COMPILE ERROR here ("the trait bound `E: GpuLz77` is not
// satisfied"), by design: running foreign CPU code per chunk in the middle of the device pipeline is what the plug-in
// direction is for — give the ORIGINAL crate's encoder a `GpuLz77Encoder` instead.
mod sealed { pub trait Sealed {} }
pub trait GpuLz77: Lz77Encode + sealed::Sealed {
    #[doc(hidden)]
    fn configure(&self, o: &mut ffi::lfx_encode_opts);
}

/// `libflate_lz77::DefaultLz77Encoder` (default.rs:14-109) — as the `E` of this crate's encoders it only carries the
/// window size and maximum match length; used directly through `Lz77Encode` it is a `GpuLz77Encoder`.
pub struct DefaultLz77Encoder {
    window_size: u16,
    max_length: u16,
    gpu: Option<GpuLz77Encoder>,
}
impl std::fmt::Debug for DefaultLz77Encoder {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "DefaultLz77Encoder {{ window_size: {}, max_length: {} }}", self.window_size, self.max_length)
    }
}
impl Default for DefaultLz77Encoder {
    fn default() -> Self { Self::new() }
}
impl DefaultLz77Encoder {
    /// default.rs:32-36 (`MAX_WINDOW_SIZE` does not fit u16 + 1: 0 stands for 32768 like the reference's wrap)
    pub fn new() -> Self { DefaultLz77Encoder { window_size: 0, max_length: MAX_LENGTH, gpu: None } }
    /// default.rs:49-57
    pub fn with_window_size(size: u16) -> Self {
        DefaultLz77Encoder { window_size: std::cmp::min(size, MAX_WINDOW_SIZE), max_length: MAX_LENGTH, gpu: None }
    }
    fn gpu(&mut self) -> &mut GpuLz77Encoder {
        if self.gpu.is_none() {
            self.gpu = Some(GpuLz77Encoder::with_window_size_and_max_length(self.window_size, self.max_length)
                .expect("libflate-amd: no usable MI355X device"));
        }
        self.gpu.as_mut().unwrap()
    }
}
impl Lz77Encode for DefaultLz77Encoder {
    fn encode<S: Sink>(&mut self, buf: &[u8], sink: S) { self.gpu().encode(buf, sink) }
    fn flush<S: Sink>(&mut self, sink: S) { self.gpu().flush(sink) }
    fn compression_level(&self) -> CompressionLevel { CompressionLevel::Balance }
    fn window_size(&self) -> u16 { if self.window_size == 0 { MAX_WINDOW_SIZE } else { self.window_size } }
}
impl sealed::Sealed for DefaultLz77Encoder {}
impl GpuLz77 for DefaultLz77Encoder {
    fn configure(&self, o: &mut ffi::lfx_encode_opts) {
        o.lz77_kind = ffi::LFX_LZ77_DEFAULT;
        o.window_size = if self.window_size == 0 { 32768 } else { self.window_size as u32 };
        o.max_length = self.max_length as u32;
    }
}

/// `libflate_lz77::DefaultLz77EncoderBuilder` (default.rs:202-249)
#[derive(Debug, Clone)]
pub struct DefaultLz77EncoderBuilder { window_size: u16, max_length: u16 }
impl Default for DefaultLz77EncoderBuilder {
    fn default() -> Self { Self::new() }
}
impl DefaultLz77EncoderBuilder {
    pub fn new() -> Self { DefaultLz77EncoderBuilder { window_size: MAX_WINDOW_SIZE, max_length: MAX_LENGTH } }
    pub fn window_size(self, window_size: u16) -> Self { DefaultLz77EncoderBuilder { window_size: std::cmp::min(window_size, MAX_WINDOW_SIZE), ..self } }
    pub fn max_length(self, max_length: u16) -> Self { DefaultLz77EncoderBuilder { max_length: std::cmp::min(max_length, MAX_LENGTH), ..self } }
    pub fn build(self) -> DefaultLz77Encoder { DefaultLz77Encoder { window_size: self.window_size, max_length: self.max_length, gpu: None } }
}

/// `libflate_lz77::NoCompressionLz77Encoder` (lib.rs:111-145): every byte a literal (still Huffman coded)
#[derive(Debug, Default)]
pub struct NoCompressionLz77Encoder;
impl NoCompressionLz77Encoder {
    pub fn new() -> Self { NoCompressionLz77Encoder }
}
impl Lz77Encode for NoCompressionLz77Encoder {
    fn encode<S: Sink>(&mut self, buf: &[u8], mut sink: S) {
        for &b in buf { sink.consume(Code::Literal(b)); }            // lib.rs:127-135 (no search: nothing to accelerate)
    }
    fn flush<S: Sink>(&mut self, _sink: S) {}
    fn compression_level(&self) -> CompressionLevel { CompressionLevel::None }
}
impl sealed::Sealed for NoCompressionLz77Encoder {}
impl GpuLz77 for NoCompressionLz77Encoder {
    fn configure(&self, o: &mut ffi::lfx_encode_opts) { o.lz77_kind = ffi::LFX_LZ77_NOCOMPRESSION; }
}
