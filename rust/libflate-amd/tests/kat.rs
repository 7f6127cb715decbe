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
fn zlib_sync_flush_issue_27() {
    // src/zlib.rs:862-902: three writes + flush, twice, FlushMode::Sync
    let writes: [&[u8]; 3] = [b"a", b"b", b"c"];
    let mut encoder = zlib::Encoder::with_options(Vec::new(), zlib::EncodeOptions::new().flush_mode(zlib::FlushMode::Sync)).unwrap();
    for _ in 0..2 {
        for w in writes.iter() { encoder.write_all(w).unwrap(); }
        encoder.flush().unwrap();
    }
    let out = encoder.finish().into_result().unwrap();
    let mut decoder = zlib::Decoder::new(&out[..]).unwrap();
    let mut buf = Vec::new();
    decoder.read_to_end(&mut buf).unwrap();
    assert_eq!(buf, b"abcabc");
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
