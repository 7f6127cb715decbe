//! `libflate::deflate` (reference `src/deflate/{mod,encode,decode}.rs`).
use crate::lz77::{DefaultLz77Encoder, GpuLz77, Lz77Stage};
use crate::{ffi, Finish, RawDecoder, RawEncoder};
use std::io;

pub const DEFAULT_BLOCK_SIZE: usize = 1024 * 1024; // encode.rs:11

/// `deflate::EncodeOptions<E>` (encode.rs:17-128).  `E` is the LZ77 stage — the reference's own parameter: ANY
/// `E: Lz77Encode + 'static` (`lz77::GpuLz77` is implemented for all of them).  This crate's `DefaultLz77Encoder` /
/// `NoCompressionLz77Encoder` run on the device with the rest of the path; any other `E` runs on the caller's side and the
/// device Huffman-codes what it emits, see lz77.rs.
#[derive(Debug)]
pub struct EncodeOptions<E = DefaultLz77Encoder>
where
    E: GpuLz77,
{
    pub(crate) block_size: usize,
    pub(crate) dynamic_huffman: bool,
    pub(crate) lz77: Option<E>,        // None = stored blocks (encode.rs:77-80)
}
impl Default for EncodeOptions<DefaultLz77Encoder> {
    fn default() -> Self { Self::new() }
}
impl EncodeOptions<DefaultLz77Encoder> {
    /// encode.rs:45-51
    pub fn new() -> Self {
        EncodeOptions { block_size: DEFAULT_BLOCK_SIZE, dynamic_huffman: true, lz77: Some(DefaultLz77Encoder::new()) }
    }
}
impl<E> EncodeOptions<E>
where
    E: GpuLz77,
{
    /// encode.rs:59-65
    pub fn with_lz77(lz77: E) -> Self { EncodeOptions { block_size: DEFAULT_BLOCK_SIZE, dynamic_huffman: true, lz77: Some(lz77) } }
    /// encode.rs:77-80
    pub fn no_compression(mut self) -> Self { self.lz77 = None; self }
    /// encode.rs:92-95
    pub fn block_size(mut self, size: usize) -> Self { self.block_size = size; self }
    /// encode.rs:107-110
    pub fn fixed_huffman_codes(mut self) -> Self { self.dynamic_huffman = false; self }

    /// → the C options and whether the whole path runs on the device for this `E` (else `E` runs on the caller's side)
    pub(crate) fn to_ffi(&self) -> (ffi::lfx_encode_opts, bool) {
        let mut o: ffi::lfx_encode_opts = unsafe { std::mem::zeroed() };
        unsafe { ffi::lfx_encode_opts_default(&mut o) };
        o.block_size = self.block_size as u64;
        o.dynamic_huffman = self.dynamic_huffman as i32;
        let on_device = match self.lz77 {
            Some(ref e) => e.configure(&mut o),
            None => { o.no_compression = 1; true }      // RawBuf never calls the Lz77Encode (encode.rs:348-383)
        };
        (o, on_device)
    }
    /// the caller-side stage of an encoder built from these options (`E` moves into it when it is not a device stage)
    pub(crate) fn into_stage(self, on_device: bool) -> Lz77Stage<E> {
        Lz77Stage { lz77: if on_device { None } else { self.lz77 }, block_size: self.block_size, original_size: 0 }
    }
}

/// `deflate::Encoder<W, E>` (encode.rs:132-249)
pub struct Encoder<W: io::Write, E = DefaultLz77Encoder> {
    raw: RawEncoder<W>,
    stage: Lz77Stage<E>,
}
impl<W: io::Write> Encoder<W, DefaultLz77Encoder> {
    /// encode.rs:156-158
    pub fn new(inner: W) -> Self {
        Self::with_options(inner, EncodeOptions::default())
    }
}
impl<W: io::Write, E: GpuLz77> Encoder<W, E> {
    /// encode.rs:182-189.  Panics when the encoder cannot be made — no usable GPU (there is no CPU fallback here), or an
    /// option outside the reference's domain; the reference constructor is infallible — `try_with_options` reports it.
    pub fn with_options(inner: W, options: EncodeOptions<E>) -> Self {
        Self::try_with_options(inner, options).unwrap_or_else(|e| panic!("libflate-amd: {}", e))
    }
    pub fn try_with_options(inner: W, options: EncodeOptions<E>) -> io::Result<Self> {
        let (o, on_device) = options.to_ffi();
        Ok(Encoder { raw: RawEncoder::new(ffi::LFX_DEFLATE, &o, inner)?, stage: options.into_stage(on_device) })
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
    fn flush(&mut self) -> io::Result<()> { self.stage.close(&mut self.raw, 1)?; self.raw.flush() }
}

/// `deflate::Decoder` (decode.rs:8-164)
pub struct Decoder<R: io::Read> {
    raw: RawDecoder<R>,
}
impl<R: io::Read> Decoder<R> {
    pub fn new(inner: R) -> Self {
        Decoder { raw: RawDecoder::new(ffi::LFX_DEFLATE, 0, inner).expect("libflate-amd: no usable MI355X device") }
    }
    pub fn as_inner_ref(&self) -> &R { self.raw.inner_ref() }
    pub fn as_inner_mut(&mut self) -> &mut R { self.raw.inner_mut() }
    pub fn into_inner(self) -> R { self.raw.into_inner() }
    pub fn unread_decoded_data(&self) -> &[u8] { self.raw.unread_decoded_data() }
    /// bytes pulled from the reader that lie behind the stream (see `RawDecoder::surplus`)
    pub fn unread_input(&self) -> &[u8] { self.raw.surplus() }
}
impl<R: io::Read> io::Read for Decoder<R> {
    fn read(&mut self, buf: &mut [u8]) -> io::Result<usize> { self.raw.read(buf) }
}
