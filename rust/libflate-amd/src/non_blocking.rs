//! `libflate::non_blocking::{deflate,zlib,gzip}::Decoder` (reference `src/non_blocking/`): the inner reader may
//! answer `io::ErrorKind::WouldBlock`; `read()` then returns the same error and can be called again — the bytes
//! seen so far stay buffered (the reference rolls a transactional bit reader back, `transaction.rs`).
use crate::{ffi, RawDecoder};
use std::io;

macro_rules! nb_decoder {
    ($name:ident, $fmt:expr, { $($extra:tt)* }) => {
        pub mod $name {
            use super::*;
            pub struct Decoder<R: io::Read> { raw: RawDecoder<R> }
            impl<R: io::Read> Decoder<R> {
                /// reads nothing yet (non_blocking/gzip.rs:64-88)
                pub fn new(inner: R) -> Self {
                    Decoder { raw: RawDecoder::new($fmt, ffi::LFX_DEC_NONBLOCKING, inner).expect("libflate-amd: no usable MI355X device") }
                }
                pub fn as_inner_ref(&self) -> &R { self.raw.inner_ref() }
                pub fn as_inner_mut(&mut self) -> &mut R { self.raw.inner_mut() }
                pub fn into_inner(self) -> R { self.raw.into_inner() }
                pub fn unread_decoded_data(&self) -> &[u8] { self.raw.unread_decoded_data() }
                pub fn unread_input(&self) -> &[u8] { self.raw.surplus() }
                $($extra)*
            }
            impl<R: io::Read> io::Read for Decoder<R> {
                fn read(&mut self, buf: &mut [u8]) -> io::Result<usize> { self.raw.read(buf) }
            }
        }
    };
}
nb_decoder!(deflate, ffi::LFX_DEFLATE, {});
nb_decoder!(zlib, ffi::LFX_ZLIB, {
    /// header of the stream; may itself answer WouldBlock (non_blocking/zlib.rs:57-71)
    pub fn header(&mut self) -> io::Result<crate::zlib::Header> { Ok(crate::zlib::Header::from_ffi(&self.raw.header()?)) }
});
nb_decoder!(gzip, ffi::LFX_GZIP, {
    /// header of the member; may itself answer WouldBlock (non_blocking/gzip.rs:98-113)
    pub fn header(&mut self) -> io::Result<crate::gzip::Header> { Ok(crate::gzip::Header::from_ffi(&self.raw.header()?)) }
});
