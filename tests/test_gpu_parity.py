"""GPU parity tests: the HIP path (through the C ABI) against the oracle and the reference's golden
vectors.  Integer / bitstream work: every comparison is bit-exact (no tolerance)."""
import gzip as pygzip
import io
import zlib as pyzlib

import numpy as np
import pytest

from golden import kat

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lfx():
    import __graft_entry__ as g
    g.build()
    import libflate_amd
    return libflate_amd


@pytest.fixture(scope="module")
def ctx(lfx):
    return lfx.Context(0)


@pytest.fixture(scope="module")
def ffi(lfx):
    from libflate_amd import _ffi
    return _ffi


@pytest.fixture(scope="module")
def synth():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import synth as s
    s.build()
    return s


def enc(ctx, ffi, fmt, data, write_size=0, writes=None, **kw):
    return ctx.encode_host(fmt, data, ffi.make_opts(**kw), ffi.make_schedule(write_size, writes))


# ------------------------------------------------------------------ reference known-answer vectors
def test_kat_encode(ctx, ffi):
    assert enc(ctx, ffi, ffi.DEFLATE, kat.HELLO) == kat.DEFLATE_HELLO                       # encode.rs:152-154
    assert enc(ctx, ffi, ffi.DEFLATE, kat.HELLO, no_compression=1) == kat.DEFLATE_HELLO_STORED  # encode.rs:178-180
    assert enc(ctx, ffi, ffi.ZLIB, kat.HELLO) == kat.ZLIB_HELLO                             # zlib.rs:547-549
    assert enc(ctx, ffi, ffi.ZLIB, kat.HELLO, no_compression=1) == kat.ZLIB_HELLO_STORED    # zlib.rs:573-575
    assert enc(ctx, ffi, ffi.GZIP, kat.HELLO, no_compression=1, mtime=123) == kat.GZIP_HELLO_STORED  # gzip.rs:800-802


@pytest.mark.parametrize("sync", [0, 2])
def test_kat_issue27_stream_api(lfx, sync):
    # zlib.rs:840-902: three writes + flush, twice, then finish — through the io::Write shaped API
    sink = io.BytesIO()
    e = lfx.zlib.Encoder.with_options(sink, lfx.zlib.EncodeOptions().flush_mode(sync))
    for _ in range(2):
        for w in kat.ISSUE27_WRITES:
            assert e.write(w) == len(w)
        e.flush()
    e.finish()
    assert sink.getvalue() == (kat.ISSUE27_ZLIB_SYNC if sync else kat.ISSUE27_ZLIB_NONE)


def test_kat_lz77_plugin(lfx):
    z = lfx.lz77.DefaultLz77Encoder.new()                                                 # lz77.rs:16-32
    sink = []
    z.encode(b"aaaaa", sink)
    z.flush(sink)
    assert sink == [lfx.lz77.Code.Literal(97), lfx.lz77.Code.Pointer(4, 1)]
    assert z.window_size() == lfx.lz77.MAX_WINDOW_SIZE and z.compression_level() == 2


def test_kat_decode(ctx, ffi, lfx):
    assert ctx.decode_host(ffi.DEFLATE, kat.DEFLATE_HELLO_FIXED)[:3] == (0, kat.HELLO, 14)    # decode.rs:28
    assert ctx.decode_host(ffi.ZLIB, kat.ZLIB_HELLO_FIXED)[:2] == (0, kat.HELLO)              # zlib.rs:708-728
    assert ctx.decode_host(ffi.GZIP, kat.GZIP_HELLO_STORED)[:2] == (0, kat.HELLO)             # gzip.rs:931-939
    both = kat.GZIP_MEMBER_HELLO_ + kat.GZIP_MEMBER_WORLD                                     # gzip.rs:1072-1083
    assert ctx.decode_host(ffi.GZIP, both, flags=ffi.DEC_MULTI)[:3] == (0, b"Hello World!", len(both))
    assert ctx.decode_host(ffi.GZIP, both)[:3] == (0, b"Hello ", len(kat.GZIP_MEMBER_HELLO_))  # gzip.rs:1216-1226
    rc, out, used, _ = ctx.decode_host(ffi.GZIP, kat.OFFSET_GZ)                               # non_blocking/gzip.rs:177
    assert (rc, out, used) == (0, kat.OFFSET_PLAIN, len(kat.OFFSET_GZ))
    assert lfx.gzip.MultiDecoder.new(both).read_to_end() == b"Hello World!"
    d = lfx.gzip.Decoder.new(both)
    assert d.read_to_end() == b"Hello " and d.consumed() == len(kat.GZIP_MEMBER_HELLO_)


def test_kat_rejects(ctx, ffi, oracle, lfx):
    vectors = [(ffi.DEFLATE, kat.TOO_LONG_BACKREF), (ffi.DEFLATE, kat.ISSUE64), (ffi.DEFLATE, kat.ISSUE3_INPUT),
               (ffi.GZIP, kat.ISSUE15_1), (ffi.GZIP, kat.ISSUE15_2), (ffi.GZIP, kat.ISSUE15_3),
               (ffi.ZLIB, kat.ISSUE71_IN), (ffi.ZLIB, kat.ISSUE82)] + [(ffi.ZLIB, d) for d in kat.ISSUES_16]
    for fmt, data in vectors:
        rc, out, used, msg = ctx.decode_host(fmt, data)
        orc, oout, oused, omsg = oracle.decode(fmt, data)
        assert rc == orc and rc != 0, (fmt, len(data))
        assert out == oout, (fmt, len(data), msg, omsg)
        assert msg.split(":")[0] == omsg.split(":")[0], (msg, omsg)
    rc, _, _, msg = ctx.decode_host(ffi.DEFLATE, kat.TOO_LONG_BACKREF)                        # decode.rs:194-212
    assert rc == ffi.E_INVALID_DATA and msg == "Too long backword reference: buffer.len=5, distance=25520"
    for d in kat.ISSUES_16:                                                                   # zlib.rs:798-837
        assert ctx.decode_host(ffi.ZLIB, d)[3][:31] == "The value of HDIST is too big: max=30, actual=32"[:31]
    rc, out, _, _ = ctx.decode_host(ffi.ZLIB, kat.ISSUE71_IN)                                 # zlib.rs:916-934
    assert rc == ffi.E_UNEXPECTED_EOF and out == kat.ISSUE71_OUT
    # the same through the io::Read shaped API: error, then unread_decoded_data()
    d = lfx.zlib.Decoder.new(kat.ISSUE71_IN)
    with pytest.raises(lfx.zlib.StreamError) as ei:
        d.read_to_end()
    assert ei.value.kind == "UnexpectedEof"
    assert ei.value.partial + d.unread_decoded_data() == kat.ISSUE71_OUT
    with pytest.raises(lfx.zlib.StreamError) as ei:                                           # zlib.rs:936-943
        lfx.zlib.Decoder.new(kat.ISSUE82)
    assert ei.value.kind == "InvalidData" and "method=0" in ei.value.message
    # issue 61: a zero-length read must not latch end of stream (gzip.rs:1249-1258)
    g = enc(ctx, ffi, ffi.GZIP, b"Hello World")
    d = lfx.gzip.Decoder.new(g)
    assert d.read(0) == b"" and d.read_to_end() == b"Hello World"


# ------------------------------------------------------------------ encode parity vs the oracle
def corpus(synth):
    rng = np.random.default_rng(42)
    return {
        "empty": b"", "one": b"a", "two": b"ab", "three": b"abc", "four": b"abcd", "aaaaa": b"aaaaa",
        "hello3": b"hello hello hello",
        "issue52": kat.ISSUE52, "test_i": kat.test_i(),
        "zeros": bytes(300000),
        "text1m": synth.text(1 << 20).tobytes(),
        "lowent1m": synth.lowent(1 << 20).tobytes(),
        "random": rng.integers(0, 256, 200000, dtype=np.uint8).tobytes(),
        "alpha4": rng.integers(0, 4, 300000, dtype=np.uint8).tobytes(),
        "ramp": kat.ramp(),
        "periodic": (b"0123456789abcdef" * 40000)[:600001],
        "trigram_far": (bytes(range(256)) * 200)[:40003],
    }


def test_encode_parity_default(ctx, ffi, oracle, synth):
    for name, data in corpus(synth).items():
        for fmt in (ffi.DEFLATE, ffi.ZLIB, ffi.GZIP):
            for ws in (0, 8192):
                got = enc(ctx, ffi, fmt, data, ws)
                assert got == oracle.encode(fmt, data, write_size=ws), (name, fmt, ws, len(got))


def test_encode_parity_options(ctx, ffi, oracle, synth):
    c = corpus(synth)
    cases = [
        ("text1m", dict(write_size=1000)), ("text1m", dict(write_size=8192, block_size=100000)),
        ("text1m", dict(write_size=300000)), ("text1m", dict(write_size=8192, dynamic_huffman=0)),
        ("text1m", dict(write_size=8192, window_size=1024)), ("text1m", dict(write_size=8192, max_length=16)),
        ("text1m", dict(write_size=8192, window_size=300, max_length=3)),
        ("test_i", dict(lz77_kind=1)), ("test_i", dict(no_compression=1)),
        ("test_i", dict(no_compression=1, block_size=1000, write_size=700)),
        ("lowent1m", dict(write_size=8192, block_size=65536)), ("zeros", dict(write_size=4096)),
        ("random", dict(write_size=8192, block_size=32768)), ("alpha4", dict(write_size=65536)),
        ("issue52", dict(write_size=100)), ("periodic", dict(write_size=8192)),
    ]
    for name, kw in cases:
        ws = kw.pop("write_size", 0)
        for fmt in (ffi.ZLIB, ffi.GZIP):
            got = enc(ctx, ffi, fmt, c[name], ws, **kw)
            assert got == oracle.encode(fmt, c[name], write_size=ws, **kw), (name, kw, ws)
    # gzip header options (gzip.rs:126-288)
    kw = dict(mtime=77, filename=b"x.txt", comment=b"hi", hcrc=1, is_text=1, os=11,
              extra=bytes([0, 0x42, 3, 0]) + b"abc")
    assert enc(ctx, ffi, ffi.GZIP, c["test_i"], 8192, **kw) == oracle.encode(oracle.GZIP, c["test_i"], 8192, **kw)


def test_encode_parity_literal_stretches_among_long_matches(ctx, ffi, oracle, synth):
    """The parse walk's instance for data with runs (picked by the match stage's sample; `parse_walk_kernel`'s cooperative finish):
    a few lanes of a wavefront walk long literal stretches while the others take one 258-byte step.  Literal stretches of every
    length 1..130 between runs, at every phase of the 52-position groups and across the 3328-position segments, runs that end
    inside / at / behind a group; text and run regions in one call (one instance for all of it); a run region behind 200 KB of
    text (the sample that picks the instance sees only part of the input)."""
    rng = np.random.default_rng(5252)
    parts = []
    for k in range(1, 131):
        parts.append(bytes([k & 0xFF]) * int(rng.integers(40, 700)))
        parts.append(rng.integers(0, 256, k, dtype=np.uint8).tobytes())            # k literals (random bytes: no matches)
    stretches = b"".join(parts)
    phased = b"".join(bytes(300 + ph) + rng.integers(0, 256, 52 + ph % 7, dtype=np.uint8).tobytes() for ph in range(0, 3400, 37))
    text = synth.text(400000).tobytes()
    low = synth.lowent(500000, seed=0x5EED00AA).tobytes()
    cases = {
        "stretches": stretches,
        "phased": phased,
        "text_then_runs": text[:200000] + low[:300000] + text[200000:] + bytes(70000) + stretches,
        "runs_then_text": low + text,
        "zeros_literal_zeros": bytes(100000) + rng.integers(0, 256, 64, dtype=np.uint8).tobytes() + bytes(100000),
    }
    for name, data in cases.items():
        for ws in (0, 8192, 3000):
            got = enc(ctx, ffi, ffi.ZLIB, data, ws)
            assert got == oracle.encode(ffi.ZLIB, data, write_size=ws), (name, ws, len(data), len(got))
            rc, out, used, msg = ctx.decode_host(ffi.ZLIB, got)
            assert rc == 0 and out == data, (name, ws, rc, msg)


def test_encode_parity_write_lists(ctx, ffi, oracle, synth, lfx):
    data = synth.text(700000).tobytes()
    rng = np.random.default_rng(8)
    for trial in range(4):
        sizes, left = [], len(data)
        while left:
            w = int(min(left, rng.integers(1, 200000)))
            sizes.append(w)
            left -= w
            if rng.integers(0, 5) == 0:
                sizes.append(None)
        for sync in (0, 2):
            e = oracle.Encoder(oracle.ZLIB, zlib_sync_flush=1 if sync else 0)
            off = 0
            for w in sizes:
                if w is None:
                    e.flush()
                else:
                    e.write(data[off:off + w]); off += w
            want = e.finish()
            got = enc(ctx, ffi, ffi.ZLIB, data, writes=sizes, zlib_flush_mode=sync)
            assert got == want, (trial, sync)
            # and through the stream API (one lfx_encoder_write per write)
            sink = io.BytesIO()
            se = lfx.zlib.Encoder.with_options(sink, lfx.zlib.EncodeOptions().flush_mode(sync))
            off = 0
            for w in sizes:
                if w is None:
                    se.flush()
                else:
                    se.write(data[off:off + w]); off += w
            se.finish()
            assert sink.getvalue() == want, (trial, sync, "stream")


def test_lz77_plugin_parity(lfx, oracle, synth):
    data = synth.text(600000).tobytes()
    for window, maxlen in ((32768, 258), (4096, 32)):
        z = lfx.lz77.DefaultLz77EncoderBuilder.new().window_size(window).max_length(maxlen).build()
        got = []
        for off in range(0, len(data), 8192):      # auto-flush at >= window*8 (default.rs:65)
            z.encode(data[off:off + 8192], got)
        z.flush(got)
        want, off, buf = [], 0, b""
        for o in range(0, len(data), 8192):
            buf += data[o:o + 8192]
            if len(buf) >= window * 8:
                want.extend(oracle.lz77_chunk(buf, window, maxlen)); buf = b""
        want.extend(oracle.lz77_chunk(buf, window, maxlen))
        assert got == [lfx.lz77.Code.from_word(w) for w in want]


# ------------------------------------------------------------------ decode parity
def test_decode_parity(ctx, ffi, oracle, synth):
    c = corpus(synth)
    for name in ("empty", "one", "hello3", "issue52", "test_i", "zeros", "text1m", "lowent1m", "random", "ramp"):
        data = c[name]
        for fmt, pyenc in ((ffi.DEFLATE, None), (ffi.ZLIB, pyzlib.compress), (ffi.GZIP, lambda d: pygzip.compress(d, mtime=0))):
            streams = [oracle.encode(fmt, data, write_size=8192), oracle.encode(fmt, data)]
            if pyenc:
                streams.append(pyenc(data))      # foreign encoder: cross-block references, fixed/stored blocks
            for s in streams:
                rc, out, used, msg = ctx.decode_host(fmt, s)
                assert (rc, used) == (0, len(s)) and out == data, (name, fmt, msg)
    # truncations and corruptions agree with the oracle (status, bytes so far, message head)
    s = oracle.encode(oracle.GZIP, c["text1m"][:200000], write_size=8192)
    rng = np.random.default_rng(1)
    for cut in [0, 5, 10, 11, 100, 5000, len(s) - 9, len(s) - 8, len(s) - 1]:
        got, want = ctx.decode_host(ffi.GZIP, s[:cut]), oracle.decode(oracle.GZIP, s[:cut])
        assert got[:2] == want[:2] and got[3].split(":")[0] == want[3].split(":")[0], cut
    for _ in range(40):
        b = bytearray(s)
        b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        got, want = ctx.decode_host(ffi.GZIP, bytes(b)), oracle.decode(oracle.GZIP, bytes(b))
        assert got[0] == want[0] and got[1] == want[1] and got[3].split(":")[0] == want[3].split(":")[0]


def test_decode_batch(ctx, ffi, oracle, synth):
    import ctypes as C
    import torch
    count = 96
    plains = [synth.text(65536, seed=0x5EED0003 + i).tobytes() for i in range(count)]
    streams = [oracle.encode(oracle.ZLIB, p) if i % 2 else pyzlib.compress(p, 6) for i, p in enumerate(plains)]
    streams[5] = streams[5][:1000]                     # truncated
    bad = bytearray(streams[7]); bad[-1] ^= 1; streams[7] = bytes(bad)   # Adler mismatch
    blob = b"".join(streams)
    in_off = np.cumsum([0] + [len(s) for s in streams[:-1]]).astype(np.uint64)
    in_len = np.array([len(s) for s in streams], dtype=np.uint64)
    out_off = (np.arange(count) * 65536).astype(np.uint64)
    out_cap = np.full(count, 65536, dtype=np.uint64)
    d_in = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
    d_out = torch.zeros(count * 65536, dtype=torch.uint8, device="cuda")
    out_len = np.zeros(count, dtype=np.uint64)
    status = np.zeros(count, dtype=np.int32)
    torch.cuda.synchronize()
    rc = ffi.lib().lfx_decode_batch_device(ctx.handle, ffi.ZLIB, count, d_in.data_ptr(), in_off.ctypes.data,
                                           in_len.ctypes.data, d_out.data_ptr(), out_off.ctypes.data,
                                           out_cap.ctypes.data, out_len.ctypes.data, status.ctypes.data)
    assert rc == 0
    host = d_out.cpu().numpy().tobytes()
    for i in range(count):
        want = oracle.decode(oracle.ZLIB, streams[i])
        assert int(status[i]) == want[0], i
        assert host[i * 65536:i * 65536 + int(out_len[i])] == want[1], i
    assert status[5] == ffi.E_UNEXPECTED_EOF and status[7] == ffi.E_INVALID_DATA


def test_device_api_and_capacity(ctx, ffi, oracle, synth):
    import torch
    data = synth.text(3 << 20)
    d_in = torch.from_numpy(data).cuda()
    sched = ffi.make_schedule(8192)
    opts = ffi.make_opts()
    bound = ffi.lib().lfx_encode_bound(data.size, None, None)
    d_out = torch.empty(bound, dtype=torch.uint8, device="cuda")
    n = ctx.encode_device(ffi.GZIP, d_in.data_ptr(), data.size, d_out.data_ptr(), bound, opts, sched)
    want = oracle.encode(oracle.GZIP, data.tobytes(), write_size=8192)
    assert d_out[:n].cpu().numpy().tobytes() == want
    # too small an output buffer is an error, never a silent truncation
    d_small = torch.empty(1 << 16, dtype=torch.uint8, device="cuda")
    with pytest.raises(ffi.LfxError) as ei:
        ctx.encode_device(ffi.GZIP, d_in.data_ptr(), data.size, d_small.data_ptr(), 1 << 16, opts, sched)
    assert ei.value.status == ffi.E_NOSPACE
    d_dec = torch.empty(data.size, dtype=torch.uint8, device="cuda")
    rc, ol, used, _ = ctx.decode_device(ffi.GZIP, d_out.data_ptr(), n, d_dec.data_ptr(), data.size)
    assert (rc, ol, used) == (0, data.size, n) and torch.equal(d_dec, d_in)
    rc, ol, _, _ = ctx.decode_device(ffi.GZIP, d_out.data_ptr(), n, d_dec.data_ptr(), 1000)
    assert rc == ffi.E_NOSPACE


def test_sharded_encode_virtual_ranks(lfx, ffi, oracle, synth):
    """SURVEY §8e with N virtual ranks on one device: per-rank prepare → (all-gather of shard infos) →
    emit at the global bit offset → OR-assembled member == what ONE encoder emits for the whole input;
    every rank can also decode its own shard alone."""
    import ctypes as C
    import torch
    from libflate_amd import sharded
    world, n = 3, 3 << 20
    datas = [synth.text(n, seed=synth.SEED_BASE + 2 + r) for r in range(world)]
    ctxs = [lfx.Context(0) for _ in range(world)]
    opts, sched = ffi.make_opts(mtime=0), ffi.make_schedule(8192)
    L = ffi.lib()
    d_ins = [torch.from_numpy(d).cuda() for d in datas]
    infos = []
    for r in range(world):
        info = ffi.ShardInfo()
        rc = L.lfx_encode_shard_prepare(ctxs[r].handle, ffi.GZIP, C.byref(opts), C.byref(sched), d_ins[r].data_ptr(),
                                        n, int(r == 0), int(r == world - 1), C.byref(info))
        assert rc == 0, ctxs[r].last_error()
        infos.append((info.total_bits, info.n_bytes, info.crc32, info.adler32))
    hdr_len = L.lfx_container_header_len(ffi.GZIP, C.byref(opts))
    start_bits, check, total_n = sharded.layout(infos, hdr_len, ffi.GZIP)
    parts, d_outs = [], []
    for r in range(world):
        cap = n + n // 4 + 65536
        d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        m = C.c_uint64(0)
        rc = L.lfx_encode_shard_emit(ctxs[r].handle, start_bits[r], check, total_n, d_out.data_ptr(), cap, C.byref(m))
        assert rc == 0, ctxs[r].last_error()
        parts.append(d_out[:m.value].cpu().numpy().tobytes())
        d_outs.append((d_out, m.value))
    member = sharded.assemble(parts, start_bits)
    whole = b"".join(d.tobytes() for d in datas)
    assert member == oracle.encode(oracle.GZIP, whole, write_size=8192)
    for r in range(world):
        d_dec = torch.zeros(n, dtype=torch.uint8, device="cuda")
        ol = C.c_uint64(0)
        sb = start_bits[r] if r == 0 else start_bits[r] & 7
        rc = L.lfx_decode_shard_device(ctxs[r].handle, d_outs[r][0].data_ptr(), d_outs[r][1], sb, infos[r][0],
                                       int(r == world - 1), d_dec.data_ptr(), n, C.byref(ol))
        assert rc == 0 and ol.value == n, ctxs[r].last_error()
        assert torch.equal(d_dec, d_ins[r])


def test_differential_random(ctx, ffi, oracle, synth):
    """Seeded differential sweep: random data kinds, sizes, write sizes and options, encode compared with the
    oracle byte for byte, the result decoded on the GPU (and by python zlib, an independent inflater)."""
    rng = np.random.default_rng(20260927)
    text = synth.text(600000).tobytes()
    low = synth.lowent(600000).tobytes()

    def sample(n):
        kind = int(rng.integers(0, 6))
        if n == 0:
            return b""
        if kind == 0:
            o = int(rng.integers(0, len(text) - n + 1)); return text[o:o + n]
        if kind == 1:
            o = int(rng.integers(0, len(low) - n + 1)); return low[o:o + n]
        if kind == 2:
            return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        if kind == 3:
            return rng.integers(0, int(rng.integers(2, 6)), n, dtype=np.uint8).tobytes()
        if kind == 4:
            period = int(rng.integers(1, 400))
            unit = rng.integers(0, 256, period, dtype=np.uint8).tobytes()
            return (unit * (n // period + 1))[:n]
        return bytes([int(rng.integers(0, 256))]) * n

    sizes = [0, 1, 2, 3, 4, 5, 63, 64, 65, 257, 258, 259, 4095, 4096, 4097, 32767, 32768, 32769, 65535, 65536,
             262143, 262144, 262145, 262147, 300001]
    for trial in range(int(__import__("os").environ.get("LFX_FUZZ_TRIALS", "60"))):
        n = int(sizes[trial % len(sizes)] if trial < len(sizes) else rng.integers(0, 400000))
        data = sample(n)
        ws = int(rng.choice([0, 1, 7, 100, 4096, 8192, 65536, 262144, 300000]))
        if ws == 1 and n > 20000:
            ws = 1000
        kw = {}
        r = int(rng.integers(0, 8))
        if r == 0: kw["dynamic_huffman"] = 0
        if r == 1: kw["no_compression"] = 1
        if r == 2: kw["lz77_kind"] = 1
        if r == 3: kw["block_size"] = int(rng.choice([1000, 65536, 100000]))
        if r == 4: kw["window_size"] = int(rng.choice([256, 1024, 4096, 32768]))
        if r == 5: kw["max_length"] = int(rng.choice([3, 4, 16, 64, 258]))
        fmt = (ffi.DEFLATE, ffi.ZLIB, ffi.GZIP)[trial % 3]
        got = enc(ctx, ffi, fmt, data, ws, **kw)
        want = oracle.encode(fmt, data, write_size=ws, **kw)
        assert got == want, (trial, n, ws, kw, fmt)
        assert pyzlib.decompress(got, {ffi.DEFLATE: -15, ffi.ZLIB: 15, ffi.GZIP: 31}[fmt]) == data, (trial, "pyzlib")
        rc, out = ctx.decode_host(fmt, got)[:2]
        assert rc == 0 and out == data, (trial, n, ws, kw, fmt, "gpu decode")
