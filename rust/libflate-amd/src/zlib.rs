//! `libflate::zlib` (reference `src/zlib.rs`).
use crate::lz77::{DefaultLz77Encoder, GpuLz77, Lz77Stage};
use crate::{ffi, Finish, RawDecoder, RawEncoder};
use std::io;

/// zlib.rs:28-58
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum CompressionLevel { Fastest = 0, Fast = 1, Default = 2, Slowest = 3 }
/// zlib.rs:184-195
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum FlushMode { None, Sync }

/// `zlib::Header` (zlib.rs:197-220); `window_size` in bytes (`Lz77WindowSize::to_u16` + 1 semantics: 256 << CINFO)
#[derive(Debug, Clone, PartialEq, Eq)]
pub struct Header { window_size: u32, compression_level: CompressionLevel }
impl Header {
    pub fn window_size(&self) -> u32 { self.window_size }
    pub fn compression_level(&self) -> CompressionLevel { self.compression_level.clone() }
    pub(crate) fn from_ffi(h: &ffi::lfx_header) -> Header {
        let l = match h.zlib_level { 0 => CompressionLevel::Fastest, 1 => CompressionLevel::Fast,
                                     2 => CompressionLevel::Default, _ => CompressionLevel::Slowest };
        Header { window_size: h.zlib_window_size, compression_level: l }
    }
}

/// `zlib::EncodeOptions<E>` (zlib.rs:414-518)
#[derive(Debug)]
pub struct EncodeOptions<E = DefaultLz77Encoder>
where
    E: GpuLz77,
{
    inner: crate::deflate::EncodeOptions<E>,
    flush_mode: FlushMode,
}
impl Default for EncodeOptions<DefaultLz77Encoder> {
    fn default() -> Self { Self::new() }
}
impl EncodeOptions<DefaultLz77Encoder> {
    pub fn new() -> Self { EncodeOptions { inner: crate::deflate::EncodeOptions::new(), flush_mode: FlushMode::None } }
}
impl<E: GpuLz77> EncodeOptions<E> {
    /// zlib.rs:441-449
    pub fn with_lz77(lz77: E) -> Self { EncodeOptions { inner: crate::deflate::EncodeOptions::with_lz77(lz77), flush_mode: FlushMode::None } }
    pub fn no_compression(mut self) -> Self { self.inner = self.inner.no_compression(); self }
    pub fn block_size(mut self, size: usize) -> Self { self.inner = self.inner.block_size(size); self }
    pub fn fixed_huffman_codes(mut self) -> Self { self.inner = self.inner.fixed_huffman_codes(); self }
    pub fn flush_mode(mut self, mode: FlushMode) -> Self { self.flush_mode = mode; self }
    fn to_ffi(&self) -> (ffi::lfx_encode_opts, bool) {
        let (mut o, on_device) = self.inner.to_ffi();
        o.zlib_flush_mode = if self.flush_mode == FlushMode::Sync { ffi::LFX_FLUSH_SYNC } else { ffi::LFX_FLUSH_NONE };
        (o, on_device)
    }
}

/// `zlib::Encoder<W, E>` (zlib.rs:522-681)
pub struct Encoder<W: io::Write, E = DefaultLz77Encoder> { raw: RawEncoder<W>, stage: Lz77Stage<E> }
impl<W: io::Write> Encoder<W, DefaultLz77Encoder> {
    /// writes the 2-byte header immediately and can fail (zlib.rs:577-585)
    pub fn new(inner: W) -> io::Result<Self> { Self::with_options(inner, EncodeOptions::default()) }
}
impl<W: io::Write, E: GpuLz77> Encoder<W, E> {
    /// zlib.rs:603-611
    pub fn with_options(inner: W, options: EncodeOptions<E>) -> io::Result<Self> {
        let (o, on_device) = options.to_ffi();
        Ok(Encoder { raw: RawEncoder::new(ffi::LFX_ZLIB, &o, inner)?, stage: options.inner.into_stage(on_device) })
    }
    pub fn finish(mut self) -> Finish<W, io::Error> {
        let closed = self.stage.close(&mut self.raw, 2);
        let (w, e) = self.raw.finish();
        Finish::new(w, closed.err().or(e))
    }
    pub fn as_inner_ref(&self) -> &W { self.raw.inner_ref() }
    pub fn as_inner_mut(&mut self) -> &mut W { self.raw.inner_mut() }
    pub fn into_inner(self) -> W { self.raw.into_inner() }
}
impl<W: io::Write, E: GpuLz77> io::Write for Encoder<W, E> {
    fn write(&mut self, buf: &[u8]) -> io::Result<usize> { self.stage.write(&mut self.raw, buf) }
    /// `FlushMode::Sync` appends the empty stored block `00 00 FF FF` (zlib.rs:666-671)
    fn flush(&mut self) -> io::Result<()> { self.stage.close(&mut self.raw, 1)?; self.raw.flush() }
}

/// `zlib::Decoder` (zlib.rs:284-410)
pub struct Decoder<R: io::Read> { raw: RawDecoder<R>, header: Header }
impl<R: io::Read> Decoder<R> {
    /// reads the header and can fail (zlib.rs:312-320)
    pub fn new(inner: R) -> io::Result<Self> {
        let mut raw = RawDecoder::new(ffi::LFX_ZLIB, 0, inner)?;
        let header = Header::from_ffi(&raw.header()?);
        Ok(Decoder { raw, header })
    }
    pub fn header(&self) -> &Header { &self.header }
    pub fn as_inner_ref(&self) -> &R { self.raw.inner_ref() }
    pub fn as_inner_mut(&mut self) -> &mut R { self.raw.inner_mut() }
    pub fn into_inner(self) -> R { self.raw.into_inner() }
    pub fn unread_decoded_data(&self) -> &[u8] { self.raw.unread_decoded_data() }
    pub fn unread_input(&self) -> &[u8] { self.raw.surplus() }
}
impl<R: io::Read> io::Read for Decoder<R> {
    fn read(&mut self, buf: &mut [u8]) -> io::Result<usize> { self.raw.read(buf) }
}
