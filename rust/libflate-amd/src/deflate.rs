//! `libflate::deflate` (reference `src/deflate/{mod,encode,decode}.rs`).
use crate::{ffi, Finish, RawDecoder, RawEncoder};
use std::io;

pub const DEFAULT_BLOCK_SIZE: usize = 1024 * 1024; // encode.rs:11

/// Which LZ77 stage runs on the GPU (`EncodeOptions::with_lz77`, encode.rs:59-65): the default encoder with a
/// window / maximum length, or `NoCompressionLz77Encoder` (libflate_lz77/src/lib.rs:111-145).
#[derive(Debug, Clone, Copy)]
pub enum Lz77 {
    Default { window_size: u16, max_length: u16 },
    NoCompression,
}

/// `deflate::EncodeOptions` (encode.rs:17-128)
#[derive(Debug, Clone)]
pub struct EncodeOptions {
    pub(crate) block_size: usize,
    pub(crate) dynamic_huffman: bool,
    pub(crate) no_compression: bool,
    pub(crate) lz77: Lz77,
}
impl Default for EncodeOptions {
    fn default() -> Self {
        EncodeOptions { block_size: DEFAULT_BLOCK_SIZE, dynamic_huffman: true, no_compression: false,
                        lz77: Lz77::Default { window_size: 32768, max_length: 258 } }
    }
}
impl EncodeOptions {
    pub fn new() -> Self { Self::default() }
    pub fn with_lz77(lz77: Lz77) -> Self { EncodeOptions { lz77, ..Self::default() } }
    pub fn no_compression(mut self) -> Self { self.no_compression = true; self }
    pub fn block_size(mut self, size: usize) -> Self { self.block_size = size; self }
    pub fn fixed_huffman_codes(mut self) -> Self { self.dynamic_huffman = false; self }

    pub(crate) fn to_ffi(&self) -> ffi::lfx_encode_opts {
        let mut o: ffi::lfx_encode_opts = unsafe { std::mem::zeroed() };
        unsafe { ffi::lfx_encode_opts_default(&mut o) };
        o.block_size = self.block_size as u64;
        o.dynamic_huffman = self.dynamic_huffman as i32;
        o.no_compression = self.no_compression as i32;
        match self.lz77 {
            Lz77::Default { window_size, max_length } => {
                o.lz77_kind = ffi::LFX_LZ77_DEFAULT;
                o.window_size = if window_size == 0 { 32768 } else { window_size as u32 };
                o.max_length = max_length as u32;
            }
            Lz77::NoCompression => o.lz77_kind = ffi::LFX_LZ77_NOCOMPRESSION,
        }
        o
    }
}

/// `deflate::Encoder` (encode.rs:136-249)
pub struct Encoder<W: io::Write> {
    raw: RawEncoder<W>,
}
impl<W: io::Write> Encoder<W> {
    pub fn new(inner: W) -> Self {
        Self::with_options(inner, EncodeOptions::default())
    }
    /// Panics when no GPU is usable (the reference constructor is infallible; there is no CPU fallback here).
    pub fn with_options(inner: W, options: EncodeOptions) -> Self {
        Self::try_with_options(inner, options).expect("libflate-amd: no usable MI355X device")
    }
    pub fn try_with_options(inner: W, options: EncodeOptions) -> io::Result<Self> {
        Ok(Encoder { raw: RawEncoder::new(ffi::LFX_DEFLATE, &options.to_ffi(), inner)? })
    }
    pub fn finish(self) -> Finish<W, io::Error> {
        let (w, e) = self.raw.finish();
        Finish::new(w, e)
    }
    pub fn as_inner_ref(&self) -> &W { self.raw.inner_ref() }
    pub fn as_inner_mut(&mut self) -> &mut W { self.raw.inner_mut() }
    pub fn into_inner(self) -> W { self.raw.into_inner() }
}
impl<W: io::Write> io::Write for Encoder<W> {
    fn write(&mut self, buf: &[u8]) -> io::Result<usize> { self.raw.write(buf) }
    fn flush(&mut self) -> io::Result<()> { self.raw.flush() }
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
