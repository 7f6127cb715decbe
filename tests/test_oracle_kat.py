"""Pins the C oracle (oracle/) against every known-answer vector the reference's own tests hold
for the DEFLATE hot path.  CPU only."""
import gzip as pygzip
import hashlib
import zlib as pyzlib

import numpy as np
import pytest

from golden import kat


def test_checksums(oracle):
    # src/checksum.rs:44-56
    assert oracle.crc32(b"abcde") == kat.CRC32_ABCDE
    assert oracle.adler32(b"abcde") == kat.ADLER32_ABCDE
    rng = np.random.default_rng(1)
    for n in (0, 1, 7, 8, 9, 5551, 5552, 5553, 70000):
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert oracle.crc32(d) == pyzlib.crc32(d)
        assert oracle.adler32(d) == pyzlib.adler32(d)
    # incremental update == one-shot (gzip.rs:890-895 updates per write)
    d = rng.integers(0, 256, 10000, dtype=np.uint8).tobytes()
    assert oracle.crc32(d[3000:], oracle.crc32(d[:3000])) == pyzlib.crc32(d)
    assert oracle.adler32(d[3000:], oracle.adler32(d[:3000])) == pyzlib.adler32(d)


def test_lz77_aaaaa(oracle):
    # src/lz77.rs:16-32
    codes = oracle.lz77_chunk(b"aaaaa")
    assert [(int(c) >> 16, int(c) & 0xFFFF) for c in codes] == kat.LZ77_AAAAA


def test_deflate_hello(oracle):
    assert oracle.encode(oracle.DEFLATE, kat.HELLO) == kat.DEFLATE_HELLO          # encode.rs:152-154
    assert oracle.encode(oracle.DEFLATE, kat.HELLO, no_compression=1) == kat.DEFLATE_HELLO_STORED


def test_zlib_hello(oracle):
    assert oracle.encode(oracle.ZLIB, kat.HELLO) == kat.ZLIB_HELLO                # zlib.rs:547-549
    assert oracle.encode(oracle.ZLIB, kat.HELLO, no_compression=1) == kat.ZLIB_HELLO_STORED


def test_gzip_stored(oracle):
    out = oracle.encode(oracle.GZIP, kat.HELLO, no_compression=1, mtime=123)      # gzip.rs:800-802
    assert out == kat.GZIP_HELLO_STORED


@pytest.mark.parametrize("sync", [0, 1])
def test_issue27(oracle, sync):
    # zlib.rs:840-902: three writes + flush, twice, then finish
    e = oracle.Encoder(oracle.ZLIB, zlib_sync_flush=sync)
    for _ in range(2):
        for w in kat.ISSUE27_WRITES:
            e.write(w)
        e.flush()
    out = e.finish()
    assert out == (kat.ISSUE27_ZLIB_SYNC if sync else kat.ISSUE27_ZLIB_NONE)
    assert pyzlib.decompress(out) == kat.ISSUE27_PLAIN


def test_secondary_pins(oracle):
    # SURVEY.md Appendix A (scratch-restatement outputs — second line of defence)
    for name, gen, ws, clen, sha in kat.SECONDARY:
        data = gen()
        out = oracle.encode(oracle.DEFLATE, data, write_size=ws)
        assert len(out) == clen, name
        assert hashlib.sha256(out).hexdigest() == sha, name
        assert pyzlib.decompress(out, -15) == data, name


def test_issue52_shrinks(oracle):
    # encode.rs:434-457
    for lim in (16031, 16032):
        assert len(oracle.encode(oracle.DEFLATE, kat.ISSUE52[:lim])) < lim


# ------------------------------------------------------------------ decoder pins
def test_decode_fixed_hello(oracle):
    rc, out, used, _ = oracle.decode(oracle.DEFLATE, kat.DEFLATE_HELLO_FIXED)     # decode.rs:28
    assert (rc, out, used) == (0, kat.HELLO, len(kat.DEFLATE_HELLO_FIXED))
    rc, out, used, _ = oracle.decode(oracle.ZLIB, kat.ZLIB_HELLO_FIXED)           # zlib.rs:708-728
    assert (rc, out) == (0, kat.HELLO)
    rc, out, used, _ = oracle.decode(oracle.GZIP, kat.GZIP_HELLO_STORED)          # gzip.rs:931-939
    assert (rc, out) == (0, kat.HELLO)


def test_decode_multi_member(oracle):
    both = kat.GZIP_MEMBER_HELLO_ + kat.GZIP_MEMBER_WORLD                         # gzip.rs:1072-1083
    rc, out, used, _ = oracle.decode(oracle.GZIP, both, multi=True)
    assert (rc, out, used) == (0, b"Hello World!", len(both))
    rc, out, used, _ = oracle.decode(oracle.GZIP, both, multi=False)              # gzip.rs:1216-1226
    assert (rc, out, used) == (0, b"Hello ", len(kat.GZIP_MEMBER_HELLO_))


def test_decode_offset_sync(oracle):
    rc, out, used, _ = oracle.decode(oracle.GZIP, kat.OFFSET_GZ)                  # non_blocking/gzip.rs:177
    assert rc == 0 and out == kat.OFFSET_PLAIN
    assert pygzip.decompress(kat.OFFSET_GZ) == kat.OFFSET_PLAIN
    assert used == len(kat.OFFSET_GZ)


def test_reject_vectors(oracle):
    rc, out, _, msg = oracle.decode(oracle.DEFLATE, kat.TOO_LONG_BACKREF)         # decode.rs:194-212
    assert rc == oracle.INVALID_DATA and msg.startswith("Too long backword reference")
    assert msg == "Too long backword reference: buffer.len=5, distance=25520"
    rc, _, _, msg = oracle.decode(oracle.DEFLATE, kat.ISSUE64)                    # decode.rs:214-220
    assert rc == oracle.INVALID_DATA and msg == "Invalid huffman coded stream"
    for data in (kat.ISSUE15_1, kat.ISSUE15_2, kat.ISSUE15_3):                    # gzip.rs:1228-1247
        rc, _, _, _ = oracle.decode(oracle.GZIP, data)
        assert rc != 0
    assert oracle.decode(oracle.GZIP, kat.ISSUE15_3)[3].startswith("Bit region conflict")
    for data in kat.ISSUES_16:                                                    # zlib.rs:798-837
        rc, _, _, msg = oracle.decode(oracle.ZLIB, data)
        assert rc == oracle.INVALID_DATA
        assert msg[:31] == "The value of HDIST is too big: max=30, actual=32"[:31]
    rc, out, _, _ = oracle.decode(oracle.ZLIB, kat.ISSUE71_IN)                    # zlib.rs:916-934
    assert rc == oracle.UNEXPECTED_EOF and out == kat.ISSUE71_OUT
    rc, _, _, msg = oracle.decode(oracle.ZLIB, kat.ISSUE82)                       # zlib.rs:936-943
    assert rc == oracle.INVALID_DATA and "method=0" in msg


def test_issue3_table_loads(oracle):
    # decode.rs:175-192: the dynamic table itself must load; the stream then runs out of input
    rc, out, _, msg = oracle.decode(oracle.DEFLATE, kat.ISSUE3_INPUT)
    assert rc != 0 and not msg.startswith(("Bit region", "The value of HDIST", "No preceding"))


def test_roundtrips(oracle):
    # deflate/mod.rs:48-64, zlib.rs:700-706 (issue 2), zlib.rs:766-796, non_blocking tests
    rng = np.random.default_rng(7)
    inputs = [kat.ramp(), kat.test_i(), b"", b"a", b"ab", b"abc", b"abcd"] + kat.ISSUE2_INPUTS
    inputs.append(rng.integers(0, 256, 100000, dtype=np.uint8).tobytes())
    inputs.append(rng.integers(0, 4, 300000, dtype=np.uint8).tobytes())
    for data in inputs:
        for fmt, dec in ((oracle.DEFLATE, lambda b: pyzlib.decompress(b, -15)),
                         (oracle.ZLIB, pyzlib.decompress), (oracle.GZIP, pygzip.decompress)):
            for ws in (0, 8192, 1000):
                enc = oracle.encode(fmt, data, write_size=ws)
                assert dec(enc) == data
                rc, out, used, msg = oracle.decode(fmt, enc)
                assert (rc, out, used) == (0, data, len(enc)), msg
    # fixed Huffman and NoCompressionLz77Encoder (lz77.rs:33-45)
    data = kat.test_i()
    for kw in (dict(dynamic_huffman=0), dict(lz77_kind=1), dict(block_size=4096),
               dict(window_size=1024), dict(max_length=16), dict(no_compression=1)):
        enc = oracle.encode(oracle.ZLIB, data, write_size=3000, **kw)
        assert pyzlib.decompress(enc) == data, kw
        assert oracle.decode(oracle.ZLIB, enc)[:2] == (0, data), kw


def test_decode_foreign_streams(oracle):
    # python-zlib-made streams (foreign block structure; cfg1 / cfg3 second set)
    data = kat.test_i() * 3
    for level in (1, 6, 9):
        rc, out, used, msg = oracle.decode(oracle.ZLIB, pyzlib.compress(data, level))
        assert (rc, out) == (0, data), msg
    g = pygzip.compress(data, mtime=0)
    assert oracle.decode(oracle.GZIP, g)[:3] == (0, data, len(g))


def test_gzip_header_options(oracle):
    out = oracle.encode(oracle.GZIP, b"hello world", mtime=5, filename=b"f.txt", comment=b"c",
                        hcrc=1, is_text=1, extra=bytes([0, 0x42, 3, 0]) + b"abc")
    rc, dec, used, msg = oracle.decode(oracle.GZIP, out)
    assert (rc, dec, used) == (0, b"hello world", len(out)), msg
    # python's gzip ignores FHCRC's value but parses the fields
    assert pygzip.decompress(out) == b"hello world"
    bad = bytearray(out)
    bad[10 + 2 + 7 + 6 + 2] ^= 1  # flip a bit of the stored CRC16
    assert oracle.decode(oracle.GZIP, bytes(bad))[0] == oracle.INVALID_DATA


def test_huffman_length_limit(oracle):
    # the limiting branch (depth > 15 / > 7) is unexercised by any reference KAT: check validity
    fib = [1, 1]
    while len(fib) < 30:
        fib.append(fib[-1] + fib[-2])
    w = oracle.huff_widths(fib, 15)
    assert w.max() == 15 and abs(sum(2.0 ** -int(x) for x in w) - 1.0) < 1e-12
    w7 = oracle.huff_widths(fib[:19], 7)
    assert w7.max() == 7 and abs(sum(2.0 ** -int(x) for x in w7) - 1.0) < 1e-12
    assert list(oracle.huff_widths([0, 5, 0], 15)) == [0, 1, 0]      # single symbol → width 1
    assert list(oracle.huff_widths([3, 5], 15)) == [1, 1]
