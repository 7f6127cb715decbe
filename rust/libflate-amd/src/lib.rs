//! `libflate-amd`: sile/libflate's public API over the MI355X-native DEFLATE hot path (hand-written HIP kernels
//! behind the C ABI of `liblfx.so`, see `include/lfx.h`).
//!
//! Switching a program from `libflate` is a matter of `use libflate_amd as libflate;` — the module paths and
//! type names are the reference's: `{deflate,zlib,gzip}::{Encoder,Decoder,EncodeOptions}`, `gzip::MultiDecoder`,
//! `gzip::{Header,HeaderBuilder}`, `zlib::{Header,FlushMode}`, `non_blocking::*::Decoder`, and
//! `lz77::GpuLz77Encoder`, an `impl libflate_lz77::Lz77Encode` for `EncodeOptions::with_lz77` of the *original*
//! crate (the plug-in point, `src/deflate/encode.rs:59-65`).
//!
//! Output is bit-exact with the reference for the same inputs, options and sequence of `write()` calls.
//! There is no CPU fallback: without a usable GPU every constructor returns an `io::Error`.
mod ffi;
pub mod deflate;
pub mod gzip;
pub mod lz77;
pub mod non_blocking;
pub mod sharded;
pub mod zlib;

use std::ffi::CStr;
use std::io;
use std::os::raw::{c_int, c_void};
use std::sync::{Arc, Mutex, OnceLock};

/// One device context (HIP stream + cached scratch).  Handles of one context take turns on the GPU.
pub struct Context(pub(crate) *mut ffi::lfx_ctx);
unsafe impl Send for Context {}
unsafe impl Sync for Context {}
impl Drop for Context {
    fn drop(&mut self) {
        unsafe { ffi::lfx_ctx_free(self.0) }
    }
}
impl Context {
    pub fn new(device: i32) -> io::Result<Arc<Context>> {
        let mut st: c_int = 0;
        let p = unsafe { ffi::lfx_ctx_new(device, &mut st) };
        if p.is_null() {
            return Err(io::Error::new(io::ErrorKind::Other, format!("no usable MI355X device {} (status {})", device, st)));
        }
        Ok(Arc::new(Context(p)))
    }
    pub(crate) fn last_error(&self) -> String {
        unsafe { CStr::from_ptr(ffi::lfx_ctx_last_error(self.0)).to_string_lossy().into_owned() }
    }
    /// Encode passes this context ran on the fallback match kernel because the default kernel's hardware assumption
    /// (ordered LDS exchanges) was seen violated: 0 on every part measured so far; a non-zero value
    /// explains a three times slower match stage.
    pub fn match_fallbacks(&self) -> u64 { unsafe { ffi::lfx_ctx_match_fallbacks(self.0) } }
}

/// Page-locked host memory (`lfx_host_alloc`): what a caller that moves large buffers through the one-shot host calls
/// should hold them in — the DMA engine then reads / writes it directly (no staging copy, the transfer runs at link rate).
/// Derefs to a byte slice.
pub struct HostBuf { p: *mut u8, n: usize }
unsafe impl Send for HostBuf {}
impl HostBuf {
    pub fn new(n: usize) -> io::Result<HostBuf> {
        let p = unsafe { ffi::lfx_host_alloc(n.max(1)) } as *mut u8;
        if p.is_null() { return Err(io::Error::new(io::ErrorKind::Other, "no page-locked memory to be had")); }
        Ok(HostBuf { p, n })
    }
}
impl Drop for HostBuf {
    fn drop(&mut self) { unsafe { ffi::lfx_host_free(self.p as *mut c_void) } }
}
impl std::ops::Deref for HostBuf {
    type Target = [u8];
    fn deref(&self) -> &[u8] { unsafe { std::slice::from_raw_parts(self.p, self.n) } }
}
impl std::ops::DerefMut for HostBuf {
    fn deref_mut(&mut self) -> &mut [u8] { unsafe { std::slice::from_raw_parts_mut(self.p, self.n) } }
}

/// The lazily created context of device 0 ("no hidden global state beyond lazily-created per-device
/// contexts").
pub fn default_context() -> io::Result<Arc<Context>> {
    static CTX: OnceLock<Mutex<Option<Arc<Context>>>> = OnceLock::new();
    let cell = CTX.get_or_init(|| Mutex::new(None));
    let mut g = cell.lock().unwrap();
    if g.is_none() {
        *g = Some(Context::new(0)?);
    }
    Ok(g.as_ref().unwrap().clone())
}

pub(crate) fn io_error(status: c_int, msg: String) -> io::Error {
    let kind = match status {
        ffi::LFX_E_INVALID_DATA => io::ErrorKind::InvalidData,
        ffi::LFX_E_UNEXPECTED_EOF => io::ErrorKind::UnexpectedEof,
        ffi::LFX_E_WOULD_BLOCK => io::ErrorKind::WouldBlock,
        ffi::LFX_E_ARG => io::ErrorKind::InvalidInput,
        _ => io::ErrorKind::Other,
    };
    io::Error::new(kind, msg)
}

// ---- callback trampolines: `user` points at the boxed inner stream ------------------------------------------
pub(crate) extern "C" fn write_tramp<W: io::Write>(user: *mut c_void, p: *const u8, n: usize) -> i64 {
    let w = unsafe { &mut *(user as *mut W) };
    let buf = unsafe { std::slice::from_raw_parts(p, n) };
    match w.write_all(buf) {
        Ok(()) => n as i64,
        Err(_) => -5,
    }
}
pub(crate) extern "C" fn flush_tramp<W: io::Write>(user: *mut c_void) -> c_int {
    let w = unsafe { &mut *(user as *mut W) };
    if w.flush().is_ok() { 0 } else { 1 }
}
pub(crate) extern "C" fn read_tramp<R: io::Read>(user: *mut c_void, p: *mut u8, cap: usize) -> i64 {
    let r = unsafe { &mut *(user as *mut R) };
    let buf = unsafe { std::slice::from_raw_parts_mut(p, cap) };
    loop {
        match r.read(buf) {
            Ok(k) => return k as i64,
            Err(ref e) if e.kind() == io::ErrorKind::Interrupted => continue,
            Err(ref e) if e.kind() == io::ErrorKind::WouldBlock => return -(ffi::LFX_E_WOULD_BLOCK as i64),
            Err(_) => return -5,
        }
    }
}

// ---- shared encoder / decoder cores ---------------------------------------------------------------------------
pub(crate) struct RawEncoder<W: io::Write> {
    pub(crate) h: *mut ffi::lfx_encoder,
    pub(crate) inner: Option<Box<W>>,    // boxed: the callbacks hold its address
    pub(crate) _ctx: Arc<Context>,
}
// (the handles are plain heap objects behind the C ABI and the context serialises every call: moving an encoder /
//  decoder to another thread is sound whenever the inner stream may move — the reference's types are Send when W / R are)
unsafe impl<W: io::Write + Send> Send for RawEncoder<W> {}
unsafe impl<R: io::Read + Send> Send for RawDecoder<R> {}

impl<W: io::Write> RawEncoder<W> {
    pub(crate) fn new(format: c_int, opts: &ffi::lfx_encode_opts, inner: W) -> io::Result<Self> {
        let ctx = default_context()?;
        let mut inner = Box::new(inner);
        let mut st: c_int = 0;
        let h = unsafe {
            ffi::lfx_encoder_new(ctx.0, format, opts, write_tramp::<W>, Some(flush_tramp::<W>),
                                 &mut *inner as *mut W as *mut c_void, &mut st)
        };
        if h.is_null() {
            return Err(io_error(st, format!("encoder construction failed: {}", ctx.last_error())));
        }
        Ok(RawEncoder { h, inner: Some(inner), _ctx: ctx })
    }
    fn err(&self, st: c_int) -> io::Error {
        io_error(st, unsafe { CStr::from_ptr(ffi::lfx_encoder_last_error(self.h)).to_string_lossy().into_owned() })
    }
    /// `io::Write::write`: ONE reference `write()` call; always consumes everything (`encode.rs:241-244`).
    pub(crate) fn write(&mut self, buf: &[u8]) -> io::Result<usize> {
        let r = unsafe { ffi::lfx_encoder_write(self.h, buf.as_ptr(), buf.len()) };
        if r < 0 { Err(self.err((-r) as c_int)) } else { Ok(r as usize) }
    }
    /// the code words a caller-side `Lz77Encode` emitted, with the bytes they stand for (`lfx_encoder_write_codes`)
    pub(crate) fn write_codes(&mut self, codes: &[u32], raw: &[u8], end_block: c_int) -> io::Result<()> {
        let st = unsafe { ffi::lfx_encoder_write_codes(self.h, codes.as_ptr(), codes.len(), raw.as_ptr(), raw.len(), end_block) };
        if st != 0 { Err(self.err(st)) } else { Ok(()) }
    }
    pub(crate) fn flush(&mut self) -> io::Result<()> {
        let st = unsafe { ffi::lfx_encoder_flush(self.h) };
        if st != 0 { Err(self.err(st)) } else { Ok(()) }
    }
    /// `Encoder::finish`: the sink keeps what was written even on error (`finish.rs:46-68`).
    pub(crate) fn finish(mut self) -> (W, Option<io::Error>) {
        let st = unsafe { ffi::lfx_encoder_finish(self.h) };
        let e = if st != 0 { Some(self.err(st)) } else { None };
        unsafe { ffi::lfx_encoder_free(self.h) };
        self.h = std::ptr::null_mut();
        (*self.inner.take().expect("inner stream"), e)
    }
    pub(crate) fn inner_ref(&self) -> &W { self.inner.as_ref().expect("inner stream") }
    pub(crate) fn inner_mut(&mut self) -> &mut W { self.inner.as_mut().expect("inner stream") }
    /// `into_inner`: drops the encoder without finishing the stream (`encode.rs:216-218`)
    pub(crate) fn into_inner(mut self) -> W {
        unsafe { ffi::lfx_encoder_free(self.h) };
        self.h = std::ptr::null_mut();
        *self.inner.take().expect("inner stream")
    }
}
impl<W: io::Write> Drop for RawEncoder<W> {
    fn drop(&mut self) {
        if !self.h.is_null() {
            unsafe { ffi::lfx_encoder_free(self.h) }
        }
    }
}

pub(crate) struct RawDecoder<R: io::Read> {
    pub(crate) h: *mut ffi::lfx_decoder,
    pub(crate) inner: Option<Box<R>>,    // boxed: the callback holds its address
    pub(crate) _ctx: Arc<Context>,
}
impl<R: io::Read> RawDecoder<R> {
    pub(crate) fn new(format: c_int, flags: u32, inner: R) -> io::Result<Self> {
        let ctx = default_context()?;
        let mut inner = Box::new(inner);
        let mut st: c_int = 0;
        let h = unsafe {
            ffi::lfx_decoder_new(ctx.0, format, flags, read_tramp::<R>, &mut *inner as *mut R as *mut c_void, &mut st)
        };
        if h.is_null() {
            return Err(io_error(st, ctx.last_error()));
        }
        Ok(RawDecoder { h, inner: Some(inner), _ctx: ctx })
    }
    fn err(&self, st: c_int) -> io::Error {
        io_error(st, unsafe { CStr::from_ptr(ffi::lfx_decoder_last_error(self.h)).to_string_lossy().into_owned() })
    }
    pub(crate) fn read(&mut self, buf: &mut [u8]) -> io::Result<usize> {
        let r = unsafe { ffi::lfx_decoder_read(self.h, buf.as_mut_ptr(), buf.len()) };
        if r < 0 { Err(self.err((-r) as c_int)) } else { Ok(r as usize) }
    }
    /// `Decoder::unread_decoded_data` (`decode.rs:68-73`)
    pub(crate) fn unread_decoded_data(&self) -> &[u8] {
        let (mut p, mut n) = (std::ptr::null(), 0usize);
        unsafe {
            ffi::lfx_decoder_unread(self.h, &mut p, &mut n);
            if n == 0 { &[] } else { std::slice::from_raw_parts(p, n) }
        }
    }
    /// Input already pulled from the reader that lies behind the decoded member(s): what a caller chains in front
    /// of the reader returned by `into_inner()` (`gzip.rs:987,1216-1226`); empty when the reader was read exactly.
    pub(crate) fn surplus(&self) -> &[u8] {
        let (mut p, mut n) = (std::ptr::null(), 0usize);
        unsafe {
            ffi::lfx_decoder_surplus(self.h, &mut p, &mut n);
            if n == 0 { &[] } else { std::slice::from_raw_parts(p, n) }
        }
    }
    pub(crate) fn consumed(&self) -> u64 {
        unsafe { ffi::lfx_decoder_consumed(self.h) }
    }
    pub(crate) fn header(&mut self) -> io::Result<ffi::lfx_header> {
        let mut h: ffi::lfx_header = unsafe { std::mem::zeroed() };
        let st = unsafe { ffi::lfx_decoder_header(self.h, &mut h) };
        if st != 0 { Err(self.err(st)) } else { Ok(h) }
    }
    pub(crate) fn into_inner(mut self) -> R {
        unsafe { ffi::lfx_decoder_free(self.h) };
        self.h = std::ptr::null_mut();
        *self.inner.take().expect("inner stream")
    }
    pub(crate) fn inner_ref(&self) -> &R { self.inner.as_ref().expect("inner stream") }
    pub(crate) fn inner_mut(&mut self) -> &mut R { self.inner.as_mut().expect("inner stream") }
}
impl<R: io::Read> Drop for RawDecoder<R> {
    fn drop(&mut self) {
        if !self.h.is_null() {
            unsafe { ffi::lfx_decoder_free(self.h) }
        }
    }
}

/// `libflate::finish::Finish` (`src/finish.rs:14-68`): the inner stream plus an optional error.
pub struct Finish<T, E> {
    value: T,
    error: Option<E>,
}
impl<T, E> Finish<T, E> {
    pub fn new(value: T, error: Option<E>) -> Self { Finish { value, error } }
    pub fn unwrap(self) -> (T, Option<E>) { (self.value, self.error) }
    pub fn into_result(self) -> Result<T, E> {
        match self.error { Some(e) => Err(e), None => Ok(self.value) }
    }
    pub fn as_result(&self) -> Result<&T, &E> {
        match self.error { Some(ref e) => Err(e), None => Ok(&self.value) }
    }
}
