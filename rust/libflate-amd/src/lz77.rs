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
        let h = unsafe { ffi::lfx_lz77_new(ctx.0, window_size as u32, max_length as u32, &mut st) };
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
// The reference runs ANY `E: Lz77Encode` and Huffman-codes what it emits (CompressBuf::{append, flush}, encode.rs:405-425).
// So do the encoders of this crate: `GpuLz77` is implemented for EVERY `E: Lz77Encode + 'static`.
//  * this module's `DefaultLz77Encoder` / `NoCompressionLz77Encoder` — the two implementations the reference ships
//    (default.rs:14-109, lib.rs:111-145), under the reference's names — only configure the device pipeline: match search,
//    parse, Huffman and bit packing all run on the GPU and `E`'s methods are never called;
//  * any other `E` (a user's own `impl Lz77Encode`) runs on the caller's side, exactly where the reference calls it, and
//    its code words go to the GPU Huffman / pack stages through `lfx_encoder_write_codes` (`Lz77Stage` below).
pub trait GpuLz77: Lz77Encode {
    /// true: the whole path runs on the device for this `E`; false: `E` runs here, the device takes its codes
    #[doc(hidden)]
    fn configure(&self, o: &mut ffi::lfx_encode_opts) -> bool;
}
impl<E: Lz77Encode + 'static> GpuLz77 for E {
    fn configure(&self, o: &mut ffi::lfx_encode_opts) -> bool {
        let any = self as &dyn std::any::Any;
        if let Some(d) = any.downcast_ref::<DefaultLz77Encoder>() {
            o.lz77_kind = ffi::LFX_LZ77_DEFAULT;
            o.window_size = d.window_size as u32;
            o.max_length = d.max_length as u32;
            true
        } else if any.is::<NoCompressionLz77Encoder>() {
            o.lz77_kind = ffi::LFX_LZ77_NOCOMPRESSION;
            true
        } else {
            // only the container header looks at these two (zlib.rs:212-220, gzip.rs:684)
            o.lz77_kind = ffi::LFX_LZ77_DEFAULT;
            o.window_size = self.window_size() as u32;
            o.lz77_level = 1 + match self.compression_level() {
                CompressionLevel::None => 0,
                CompressionLevel::Fast => 1,
                CompressionLevel::Balance => 2,
                CompressionLevel::Best => 3,
            };
            false
        }
    }
}

/// `Vec<u32>` as a `Sink`: code words as the C ABI takes them, (val << 16) | dist, dist == 0 for a literal
#[derive(Default)]
pub(crate) struct CodeWords(pub(crate) Vec<u32>);
impl Sink for CodeWords {
    fn consume(&mut self, code: Code) {
        self.0.push(match code {
            Code::Literal(b) => (b as u32) << 16,
            Code::Pointer { length, backward_distance } => (length as u32) << 16 | backward_distance as u32,
        });
    }
}

/// What `Block` / `CompressBuf` do around a caller-side `E` (encode.rs:277-303, 386-426): `lz77` is `Some` only for an
/// `E` the device pipeline does not implement itself.
pub(crate) struct Lz77Stage<E> {
    pub(crate) lz77: Option<E>,
    pub(crate) block_size: usize,
    pub(crate) original_size: usize,
}
impl<E: Lz77Encode> Lz77Stage<E> {
    /// `Block::write`: append, then close blocks while `block_size` is reached (encode.rs:277-286)
    pub(crate) fn write<W: io::Write>(&mut self, raw: &mut crate::RawEncoder<W>, buf: &[u8]) -> io::Result<usize> {
        let lz77 = match self.lz77 { Some(ref mut e) => e, None => return raw.write(buf) };
        let mut sink = CodeWords::default();
        lz77.encode(buf, &mut sink);                                        // CompressBuf::append encode.rs:405-408
        raw.write_codes(&sink.0, buf, 0)?;
        self.original_size += buf.len();
        while self.original_size >= self.block_size {
            sink.0.clear();
            lz77.flush(&mut sink);                                          // CompressBuf::flush encode.rs:416
            raw.write_codes(&sink.0, &[], 1)?;
            self.original_size = 0;
        }
        Ok(buf.len())
    }
    /// the block closes: `end_block` 1 = `Encoder::flush` (encode.rs:245-248), 2 = `Block::finish` (encode.rs:296-303)
    pub(crate) fn close<W: io::Write>(&mut self, raw: &mut crate::RawEncoder<W>, end_block: c_int) -> io::Result<()> {
        if let Some(ref mut lz77) = self.lz77 {
            let mut sink = CodeWords::default();
            lz77.flush(&mut sink);
            raw.write_codes(&sink.0, &[], end_block)?;
            self.original_size = 0;
        }
        Ok(())
    }
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
    /// default.rs:32-36
    pub fn new() -> Self { DefaultLz77Encoder { window_size: MAX_WINDOW_SIZE, max_length: MAX_LENGTH, gpu: None } }
    /// default.rs:49-57
    pub fn with_window_size(size: u16) -> Self {
        DefaultLz77Encoder { window_size: std::cmp::min(size, MAX_WINDOW_SIZE), max_length: MAX_LENGTH, gpu: None }
    }
    fn gpu(&mut self) -> &mut GpuLz77Encoder {
        if self.gpu.is_none() {
            self.gpu = Some(GpuLz77Encoder::with_window_size_and_max_length(self.window_size, self.max_length)
                .unwrap_or_else(|e| panic!("libflate-amd: {}", e)));
        }
        self.gpu.as_mut().unwrap()
    }
}
impl Lz77Encode for DefaultLz77Encoder {
    fn encode<S: Sink>(&mut self, buf: &[u8], sink: S) { self.gpu().encode(buf, sink) }
    fn flush<S: Sink>(&mut self, sink: S) { self.gpu().flush(sink) }
    fn compression_level(&self) -> CompressionLevel { CompressionLevel::Balance }
    fn window_size(&self) -> u16 { self.window_size }
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
