//! The reference's own doctests / unit tests, transcribed against this crate (same bytes expected).
//! Needs an MI355X and liblfx.so (LFX_LIB_DIR); not runnable in the build container (no Rust toolchain there) —
//! the same vectors run through the C ABI in tests/c/shim_abi.c and through ctypes in tests/test_gpu_parity.py.
use libflate_amd::{deflate, gzip, zlib};
use std::io::{Read, Write};

#[test]
fn deflate_encode_hello() {
    // src/deflate/encode.rs:144-155
    let mut encoder = deflate::Encoder::new(Vec::new());
    encoder.write_all(b"Hello World!").unwrap();
    assert_eq!(encoder.finish().into_result().unwrap(),
               [5, 192, 49, 13, 0, 0, 8, 3, 65, 43, 224, 6, 7, 24, 128, 237, 147, 38, 245, 63, 244, 230, 65, 181, 50, 215, 1]);
}

#[test]
fn deflate_encode_stored() {
    // src/deflate/encode.rs:170-181
    let options = deflate::EncodeOptions::new().no_compression();
    let mut encoder = deflate::Encoder::with_options(Vec::new(), options);
    encoder.write_all(b"Hello World!").unwrap();
    assert_eq!(encoder.finish().into_result().unwrap(),
               [1, 12, 0, 243, 255, 72, 101, 108, 108, 111, 32, 87, 111, 114, 108, 100, 33]);
}

#[test]
fn zlib_encode_hello() {
    // src/zlib.rs:540-550
    let mut encoder = zlib::Encoder::new(Vec::new()).unwrap();
    encoder.write_all(b"Hello World!").unwrap();
    assert_eq!(encoder.finish().into_result().unwrap(),
               vec![120, 156, 5, 192, 49, 13, 0, 0, 8, 3, 65, 43, 224, 6, 7, 24, 128, 237, 147, 38, 245, 63, 244, 230, 65, 181, 50,
                    215, 1, 28, 73, 4, 62]);
}

#[test]
fn zlib_issue_27_both_flush_modes() {
    // src/zlib.rs:838-902, byte for byte: three writes + flush, twice — FlushMode::None, then FlushMode::Sync
    let writes = ["fooooooooooooooooo", "bar", "baz"];
    let mut encoder = zlib::Encoder::new(Vec::new()).unwrap();
    for _ in 0..2 {
        for string in &writes { encoder.write(string.as_bytes()).expect("Write failed"); }
        encoder.flush().expect("Flush failed");
    }
    let finished = encoder.finish().unwrap();
    let expected = vec![
        120, 156, // header
        92, 192, 161, 17, 0, 0, 0, 1, 192, 89, 9, 170, 59, 209, 244, 186, 151, 31, 17, 162,
        227, 2, 14, 141, 0, 0, 0, 8, 0, 206, 74, 80, 221, 137, 166, 215, 189, 252, 136, 16, 93,
        1, 112, 32, 0, 0, 0, 0, 0, 228, 255, 26, 246, 95, 20, 111,
    ];
    assert_eq!(finished.0, expected);
    let mut output = Vec::new();
    zlib::Decoder::new(&finished.0[..]).unwrap().read_to_end(&mut output).unwrap();
    assert_eq!(output, "fooooooooooooooooobarbazfooooooooooooooooobarbaz".as_bytes());

    let mut encoder = zlib::Encoder::with_options(Vec::new(), zlib::EncodeOptions::new().flush_mode(zlib::FlushMode::Sync)).unwrap();
    for _ in 0..2 {
        for string in &writes { encoder.write(string.as_bytes()).expect("Write failed"); }
        encoder.flush().expect("Flush failed");
    }
    let finished = encoder.finish().unwrap();
    let expected = vec![
        120, 156, // header
        92, 192, 161, 17, 0, 0, 0, 1, 192, 89, 9, 170, 59, 209, 244, 186, 151, 31, 17, 162, 3,
        0, 0, 255, 255, // sync bytes
        92, 192, 161, 17, 0, 0, 0, 1, 192, 89, 9, 170, 59, 209, 244, 186, 151, 31, 17, 162, 3,
        0, 0, 255, 255, // sync bytes
        5, 192, 129, 0, 0, 0, 0, 0, 144, 255, 107, 0, 246, 95, 20, 111,
    ];
    assert_eq!(finished.0, expected);
    let mut output = Vec::new();
    zlib::Decoder::new(&finished.0[..]).unwrap().read_to_end(&mut output).unwrap();
    assert_eq!(output, "fooooooooooooooooobarbazfooooooooooooooooobarbaz".as_bytes());
}

#[test]
fn encode_options_carry_the_lz77_type_parameter() {
    // src/deflate/encode.rs:17,59-65,132; src/gzip.rs:639,754; src/zlib.rs:414,522: EncodeOptions<E> / Encoder<W, E>
    use libflate_amd::lz77::{DefaultLz77Encoder, DefaultLz77EncoderBuilder, NoCompressionLz77Encoder};
    let options: deflate::EncodeOptions<DefaultLz77Encoder> =
        deflate::EncodeOptions::with_lz77(DefaultLz77EncoderBuilder::new().window_size(1024).max_length(20).build());
    let mut encoder: deflate::Encoder<Vec<u8>, DefaultLz77Encoder> = deflate::Encoder::with_options(Vec::new(), options);
    encoder.write_all(b"abcabcabcabcabcabcabcabc").unwrap();
    let out = encoder.finish().into_result().unwrap();
    let mut back = Vec::new();
    deflate::Decoder::new(&out[..]).read_to_end(&mut back).unwrap();
    assert_eq!(back, b"abcabcabcabcabcabcabcabc");
    // every byte a literal (libflate_lz77/src/lib.rs:111-145), still Huffman coded
    let options = gzip::EncodeOptions::with_lz77(NoCompressionLz77Encoder::new());
    let mut encoder: gzip::Encoder<Vec<u8>, NoCompressionLz77Encoder> = gzip::Encoder::with_options(Vec::new(), options).unwrap();
    encoder.write_all(b"aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa").unwrap();
    let out = encoder.finish().into_result().unwrap();
    let mut back = Vec::new();
    gzip::Decoder::new(&out[..]).unwrap().read_to_end(&mut back).unwrap();
    assert_eq!(back.len(), 32);
    // the default parameter: `Encoder<W>` is `Encoder<W, DefaultLz77Encoder>`
    let _e: zlib::Encoder<Vec<u8>> = zlib::Encoder::new(Vec::new()).unwrap();
}

#[test]
fn zlib_decode_hello() {
    // src/zlib.rs:700-730
    let encoded = [120, 156, 243, 72, 205, 201, 201, 87, 8, 207, 47, 202, 73, 81, 4, 0, 28, 73, 4, 62];
    let mut decoder = zlib::Decoder::new(&encoded[..]).unwrap();
    let mut buf = Vec::new();
    decoder.read_to_end(&mut buf).unwrap();
    assert_eq!(buf, b"Hello World!");
}

#[test]
fn gzip_encode_stored_with_mtime() {
    // src/gzip.rs:792-803
    let header = gzip::HeaderBuilder::new().modification_time(123).finish();
    let options = gzip::EncodeOptions::new().no_compression().header(header);
    let mut encoder = gzip::Encoder::with_options(Vec::new(), options).unwrap();
    encoder.write_all(b"Hello World!").unwrap();
    assert_eq!(encoder.finish().into_result().unwrap(),
               &[31, 139, 8, 0, 123, 0, 0, 0, 0, 3, 1, 12, 0, 243, 255, 72, 101, 108, 108, 111, 32, 87, 111, 114, 108, 100, 33, 163,
                 28, 41, 28, 12, 0, 0, 0][..]);
}

#[test]
fn gzip_multi_member() {
    // src/gzip.rs:1072-1083 and 1216-1226
    let a = [31, 139, 8, 0, 51, 206, 75, 90, 0, 3, 5, 128, 49, 9, 0, 0, 0, 194, 170, 24, 199, 34, 126, 3, 251, 127, 163, 131, 71, 192,
             252, 45, 234, 6, 0, 0, 0];
    let b = [31, 139, 8, 0, 227, 207, 75, 90, 0, 3, 5, 128, 49, 9, 0, 0, 0, 194, 178, 152, 202, 2, 158, 130, 96, 255, 99, 120, 111, 4,
             222, 157, 40, 118, 6, 0, 0, 0];
    let both: Vec<u8> = a.iter().chain(b.iter()).cloned().collect();
    let mut decoder = gzip::MultiDecoder::new(&both[..]).unwrap();
    assert_eq!(decoder.header().modification_time(), 0x5A4BCE33);          // `&Header` of the member being read (gzip.rs:1106)
    let mut buf = Vec::new();
    decoder.read_to_end(&mut buf).unwrap();
    assert_eq!(buf, b"Hello World!");
    let mut decoder = gzip::Decoder::new(&both[..]).unwrap();
    let mut buf = Vec::new();
    decoder.read_to_end(&mut buf).unwrap();
    assert_eq!(buf, b"Hello ");
    assert_eq!(decoder.consumed() as usize, a.len());
    assert_eq!(decoder.unread_input(), &b[..]);
}

#[test]
fn lz77_plugin_aaaaa() {
    // libflate's src/lz77.rs:16-32 through the ORIGINAL crate's plug-in point
    use libflate_lz77::{Code, Lz77Encode};
    let mut encoder = libflate_amd::lz77::GpuLz77Encoder::new().unwrap();
    let mut codes = Vec::new();
    encoder.encode(b"aaaaa", &mut codes);
    encoder.flush(&mut codes);
    assert_eq!(codes[0], Code::Literal(97));
    assert_eq!(codes[1], Code::Pointer { length: 4, backward_distance: 1 });
}


/// A user-defined `Lz77Encode` (EncodeOptions::with_lz77(E), src/deflate/encode.rs:59-65): it runs on the caller's side and
/// the device Huffman-codes what it emits.  "Every byte a literal, CompressionLevel::None" must give the stream the device
/// pipeline makes for the reference's own NoCompressionLz77Encoder (libflate_lz77/src/lib.rs:111-145).
struct EveryByteALiteral;
impl libflate_amd::lz77::Lz77Encode for EveryByteALiteral {
    fn encode<S: libflate_amd::lz77::Sink>(&mut self, buf: &[u8], mut sink: S) {
        for &b in buf { sink.consume(libflate_amd::lz77::Code::Literal(b)); }
    }
    fn flush<S: libflate_amd::lz77::Sink>(&mut self, _sink: S) {}
    fn compression_level(&self) -> libflate_amd::lz77::CompressionLevel { libflate_amd::lz77::CompressionLevel::None }
}

#[test]
fn user_defined_lz77_encoder() {
    let data: Vec<u8> = (0..300_000u32).map(|i| (i % 251) as u8 ^ (i / 7 % 13) as u8).collect();
    let mut a = zlib::Encoder::with_options(Vec::new(), zlib::EncodeOptions::with_lz77(EveryByteALiteral).block_size(100_000)).unwrap();
    let mut b = zlib::Encoder::with_options(Vec::new(), zlib::EncodeOptions::with_lz77(libflate_amd::lz77::NoCompressionLz77Encoder::new()).block_size(100_000)).unwrap();
    for chunk in data.chunks(8192) { a.write_all(chunk).unwrap(); b.write_all(chunk).unwrap(); }
    a.flush().unwrap();
    b.flush().unwrap();
    let (a, b) = (a.finish().into_result().unwrap(), b.finish().into_result().unwrap());
    assert_eq!(a, b);
    let mut out = Vec::new();
    zlib::Decoder::new(&a[..]).unwrap().read_to_end(&mut out).unwrap();
    assert_eq!(out, data);
}
