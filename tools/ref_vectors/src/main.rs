//! Writes, for every `<name>.bin` under `<dir>/inputs/`, what the REAL libflate emits for it:
//!     <dir>/<name>.<format>.<schedule>.bin      format: deflate | zlib | gzip;  schedule: s1 | s8k
//! s1 = one `write_all` (one LZ77 chunk, one block), s8k = 8192-byte `write` calls (what `io::copy` issues,
//! examples/flate.rs:52) — the write schedule decides the bytes (SURVEY.md "fact 2").  gzip headers carry mtime 0 so that
//! the files are reproducible.  `tests/test_ref_vectors.py` compares the oracle's output with each file.
use std::fs;
use std::io::Write;
use std::path::Path;

use libflate::{deflate, gzip, zlib};

fn feed<W: Write>(w: &mut W, data: &[u8], write_size: usize) {
    if write_size == 0 {
        w.write_all(data).unwrap();
    } else {
        for c in data.chunks(write_size) {
            // one write() call per chunk: libflate's Encoder::write always consumes the whole slice (encode.rs:243)
            assert_eq!(w.write(c).unwrap(), c.len());
        }
    }
}

fn main() {
    let dir = std::env::args().nth(1).expect("usage: lfx-ref-vectors <tests/golden/ref>");
    let dir = Path::new(&dir);
    let mut names: Vec<_> = fs::read_dir(dir.join("inputs")).expect("run tools/make_ref_inputs.py first")
        .filter_map(|e| e.ok()).map(|e| e.path()).filter(|p| p.extension().map_or(false, |x| x == "bin")).collect();
    names.sort();
    for path in names {
        let name = path.file_stem().unwrap().to_string_lossy().into_owned();
        let data = fs::read(&path).unwrap();
        for (sched, ws) in [("s1", 0usize), ("s8k", 8192usize)] {
            let mut e = deflate::Encoder::new(Vec::new());
            feed(&mut e, &data, ws);
            fs::write(dir.join(format!("{name}.deflate.{sched}.bin")), e.finish().into_result().unwrap()).unwrap();

            let mut e = zlib::Encoder::new(Vec::new()).unwrap();
            feed(&mut e, &data, ws);
            fs::write(dir.join(format!("{name}.zlib.{sched}.bin")), e.finish().into_result().unwrap()).unwrap();

            let header = gzip::HeaderBuilder::new().modification_time(0).finish();
            let opts = gzip::EncodeOptions::new().header(header);
            let mut e = gzip::Encoder::with_options(Vec::new(), opts).unwrap();
            feed(&mut e, &data, ws);
            fs::write(dir.join(format!("{name}.gzip.{sched}.bin")), e.finish().into_result().unwrap()).unwrap();
        }
        println!("{name}: {} bytes", data.len());
    }
}
