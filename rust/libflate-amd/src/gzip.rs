//! `libflate::gzip` (reference `src/gzip.rs`).
use crate::lz77::{DefaultLz77Encoder, GpuLz77, Lz77Stage};
use crate::{ffi, Finish, RawDecoder, RawEncoder};
use std::ffi::{CStr, CString};
use std::io;

/// gzip.rs:58-92 (XFL)
#[derive(Debug, Clone, PartialEq, Eq)]
pub enum CompressionLevel { Fastest, Slowest, Unknown }
/// gzip.rs:545-636 — carried as the raw OS byte (3 = Unix, the encoder's default)
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub struct Os(pub u8);
impl Os { pub const UNIX: Os = Os(3); }

/// gzip.rs:503-541
#[derive(Debug, Clone, PartialEq, Eq)]
pub struct ExtraSubField { pub id: [u8; 2], pub data: Vec<u8> }
/// gzip.rs:466-501
#[derive(Debug, Clone, PartialEq, Eq)]
pub struct ExtraField { pub subfields: Vec<ExtraSubField> }
impl ExtraField {
    fn to_bytes(&self) -> Vec<u8> {
        let mut b = Vec::new();
        for f in &self.subfields {
            b.extend_from_slice(&f.id);
            b.extend_from_slice(&(f.data.len() as u16).to_le_bytes());
            b.extend_from_slice(&f.data);
        }
        b
    }
    fn from_bytes(mut b: &[u8]) -> ExtraField {
        let mut subfields = Vec::new();
        while b.len() >= 4 {
            let n = u16::from_le_bytes([b[2], b[3]]) as usize;
            if b.len() < 4 + n { break; }
            subfields.push(ExtraSubField { id: [b[0], b[1]], data: b[4..4 + n].to_vec() });
            b = &b[4 + n..];
        }
        ExtraField { subfields }
    }
}

/// `gzip::Header` (gzip.rs:292-341)
#[derive(Debug, Clone, PartialEq, Eq)]
pub struct Header {
    modification_time: u32,
    compression_level: CompressionLevel,
    os: Os,
    is_text: bool,
    is_verified: bool,
    extra_field: Option<ExtraField>,
    filename: Option<CString>,
    comment: Option<CString>,
}
impl Header {
    pub fn modification_time(&self) -> u32 { self.modification_time }
    pub fn compression_level(&self) -> CompressionLevel { self.compression_level.clone() }
    pub fn os(&self) -> Os { self.os }
    pub fn is_text(&self) -> bool { self.is_text }
    pub fn is_verified(&self) -> bool { self.is_verified }
    pub fn extra_field(&self) -> Option<&ExtraField> { self.extra_field.as_ref() }
    pub fn filename(&self) -> Option<&CString> { self.filename.as_ref() }
    pub fn comment(&self) -> Option<&CString> { self.comment.as_ref() }
    pub(crate) fn from_ffi(h: &ffi::lfx_header) -> Header {
        let cs = |p: *const std::os::raw::c_char| if p.is_null() { None } else { Some(unsafe { CStr::from_ptr(p) }.to_owned()) };
        Header {
            modification_time: h.mtime,
            compression_level: match h.xfl { 2 => CompressionLevel::Slowest, 4 => CompressionLevel::Fastest, _ => CompressionLevel::Unknown },
            os: Os(h.os),
            is_text: h.is_text != 0,
            is_verified: h.is_verified != 0,
            extra_field: if h.has_extra != 0 {
                Some(ExtraField::from_bytes(unsafe { std::slice::from_raw_parts(h.extra, h.extra_len as usize) }))
            } else { None },
            // (a present-but-empty FNAME / FCOMMENT is Some("") in the reference, gzip.rs:415-431: lfx_header hands out a
            //  pointer to the NUL in that case and NULL only when the flag bit is clear)
            filename: cs(h.filename),
            comment: cs(h.comment),
        }
    }
}

/// `gzip::HeaderBuilder` (gzip.rs:126-288).  `modification_time` defaults to "now" like the reference
/// (gzip.rs:148-151); set it for reproducible bytes.
#[derive(Debug, Clone)]
pub struct HeaderBuilder { header: Header }
impl Default for HeaderBuilder { fn default() -> Self { Self::new() } }
impl HeaderBuilder {
    pub fn new() -> Self {
        let now = std::time::SystemTime::now().duration_since(std::time::UNIX_EPOCH).map(|d| d.as_secs() as u32).unwrap_or(0);
        HeaderBuilder { header: Header { modification_time: now, compression_level: CompressionLevel::Unknown, os: Os::UNIX,
                                         is_text: false, is_verified: false, extra_field: None, filename: None, comment: None } }
    }
    pub fn modification_time(&mut self, t: u32) -> &mut Self { self.header.modification_time = t; self }
    pub fn os(&mut self, os: Os) -> &mut Self { self.header.os = os; self }
    pub fn text(&mut self) -> &mut Self { self.header.is_text = true; self }
    pub fn verify(&mut self) -> &mut Self { self.header.is_verified = true; self }
    pub fn extra_field(&mut self, extra: ExtraField) -> &mut Self { self.header.extra_field = Some(extra); self }
    pub fn filename(&mut self, filename: CString) -> &mut Self { self.header.filename = Some(filename); self }
    pub fn comment(&mut self, comment: CString) -> &mut Self { self.header.comment = Some(comment); self }
    pub fn finish(&self) -> Header { self.header.clone() }
}

/// `gzip::EncodeOptions<E>` (gzip.rs:639-751)
#[derive(Debug)]
pub struct EncodeOptions<E = DefaultLz77Encoder>
where
    E: GpuLz77,
{
    inner: crate::deflate::EncodeOptions<E>,
    header: Header,             // its compression_level IS the XFL byte the encoder writes (gzip.rs:84-92,368-389)
}
/// gzip.rs:84-92: `From<lz77::CompressionLevel> for CompressionLevel`
fn level_of(l: crate::lz77::CompressionLevel) -> CompressionLevel {
    match l {
        crate::lz77::CompressionLevel::Fast => CompressionLevel::Fastest,
        crate::lz77::CompressionLevel::Best => CompressionLevel::Slowest,
        _ => CompressionLevel::Unknown,
    }
}
impl Default for EncodeOptions<DefaultLz77Encoder> {
    fn default() -> Self { Self::new() }
}
impl EncodeOptions<DefaultLz77Encoder> {
    pub fn new() -> Self { Self::with_lz77(DefaultLz77Encoder::new()) }
}
impl<E: GpuLz77> EncodeOptions<E> {
    /// gzip.rs:667-672
    pub fn with_lz77(lz77: E) -> Self {
        let mut header = HeaderBuilder::new().finish();
        header.compression_level = level_of(lz77.compression_level());      // gzip.rs:684
        EncodeOptions { inner: crate::deflate::EncodeOptions::with_lz77(lz77), header }
    }
    /// gzip.rs:700-705: stored blocks; the header's level goes back to Unknown
    pub fn no_compression(mut self) -> Self { self.inner = self.inner.no_compression(); self.header.compression_level = CompressionLevel::Unknown; self }
    /// gzip.rs:717-720: the header REPLACES the one the options held — its own compression_level included (a header cloned
    /// from a Decoder keeps its Fastest / Slowest)
    pub fn header(mut self, header: Header) -> Self { self.header = header; self }
    pub fn block_size(mut self, size: usize) -> Self { self.inner = self.inner.block_size(size); self }
    pub fn fixed_huffman_codes(mut self) -> Self { self.inner = self.inner.fixed_huffman_codes(); self }
}

/// `gzip::Encoder<W, E>` (gzip.rs:754-908)
pub struct Encoder<W: io::Write, E = DefaultLz77Encoder> { raw: RawEncoder<W>, header: Header, stage: Lz77Stage<E> }
impl<W: io::Write> Encoder<W, DefaultLz77Encoder> {
    /// writes the header immediately and can fail (gzip.rs:804-812)
    pub fn new(inner: W) -> io::Result<Self> { Self::with_options(inner, EncodeOptions::default()) }
}
impl<W: io::Write, E: GpuLz77> Encoder<W, E> {
    /// gzip.rs:830-838
    pub fn with_options(inner: W, options: EncodeOptions<E>) -> io::Result<Self> {
        let (mut o, on_device) = options.inner.to_ffi();
        let h = &options.header;
        let extra = h.extra_field.as_ref().map(|e| e.to_bytes());
        o.mtime = h.modification_time;
        o.os = h.os.0;
        o.is_text = h.is_text as u8;
        o.hcrc = h.is_verified as u8;
        // XFL is the level of the header the options hold (gzip.rs:368-389): with_lz77 set it from E::compression_level
        // (gzip.rs:684), header() replaced it with that header's own, no_compression() reset it (gzip.rs:703)
        o.lz77_level = 1 + match h.compression_level {
            CompressionLevel::Fastest => ffi::LFX_LEVEL_FAST,
            CompressionLevel::Slowest => ffi::LFX_LEVEL_BEST,
            CompressionLevel::Unknown => ffi::LFX_LEVEL_BALANCE,
        } as u8;
        if let Some(ref e) = extra { o.extra = e.as_ptr(); o.extra_len = e.len() as u32; }
        if let Some(ref f) = h.filename { o.filename = f.as_ptr(); }
        if let Some(ref c) = h.comment { o.comment = c.as_ptr(); }
        // (lfx_encoder_new copies the strings and the extra field before it returns)
        let raw = RawEncoder::new(ffi::LFX_GZIP, &o, inner)?;
        let header = options.header.clone();
        Ok(Encoder { raw, header, stage: options.inner.into_stage(on_device) })
    }
    pub fn header(&self) -> &Header { &self.header }
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

/// `gzip::Decoder` (gzip.rs:912-1047): one member; bytes behind its trailer are not decoded
pub struct Decoder<R: io::Read> { raw: RawDecoder<R>, header: Header }
impl<R: io::Read> Decoder<R> {
    /// reads the header and can fail (gzip.rs:941-944)
    pub fn new(inner: R) -> io::Result<Self> {
        let mut raw = RawDecoder::new(ffi::LFX_GZIP, 0, inner)?;
        let header = Header::from_ffi(&raw.header()?);
        Ok(Decoder { raw, header })
    }
    pub fn header(&self) -> &Header { &self.header }
    pub fn as_inner_ref(&self) -> &R { self.raw.inner_ref() }
    pub fn as_inner_mut(&mut self) -> &mut R { self.raw.inner_mut() }
    /// The reader is positioned behind everything pulled so far; `unread_input()` holds the bytes that lie behind
    /// the member's trailer (chain them in front: `io::Cursor::new(d.unread_input().to_vec()).chain(d.into_inner())`
    /// reproduces gzip.rs:1216-1226 for readers that cannot be rewound).
    pub fn into_inner(self) -> R { self.raw.into_inner() }
    pub fn unread_decoded_data(&self) -> &[u8] { self.raw.unread_decoded_data() }
    pub fn unread_input(&self) -> &[u8] { self.raw.surplus() }
    /// bytes of the reader that belong to the member
    pub fn consumed(&self) -> u64 { self.raw.consumed() }
}
impl<R: io::Read> io::Read for Decoder<R> {
    fn read(&mut self, buf: &mut [u8]) -> io::Result<usize> { self.raw.read(buf) }
}

/// `gzip::MultiDecoder` (gzip.rs:1052-1167): all members of a concatenated stream
pub struct MultiDecoder<R: io::Read> { raw: RawDecoder<R>, header: Header }
impl<R: io::Read> MultiDecoder<R> {
    /// reads the first member's header and can fail (gzip.rs:1089-1095)
    pub fn new(inner: R) -> io::Result<Self> {
        let mut raw = RawDecoder::new(ffi::LFX_GZIP, ffi::LFX_DEC_MULTI, inner)?;
        let header = Header::from_ffi(&raw.header()?);
        Ok(MultiDecoder { raw, header })
    }
    /// header of the member being read (gzip.rs:1106: `&Header`; refreshed by `read` when it crosses into a member)
    pub fn header(&self) -> &Header { &self.header }
    pub fn as_inner_ref(&self) -> &R { self.raw.inner_ref() }
    pub fn as_inner_mut(&mut self) -> &mut R { self.raw.inner_mut() }
    pub fn into_inner(self) -> R { self.raw.into_inner() }
}
impl<R: io::Read> io::Read for MultiDecoder<R> {
    fn read(&mut self, buf: &mut [u8]) -> io::Result<usize> {
        let n = self.raw.read(buf)?;
        if n != 0 {
            if let Ok(h) = self.raw.header() { self.header = Header::from_ffi(&h); }     // (the member these bytes came from)
        }
        Ok(n)
    }
}
