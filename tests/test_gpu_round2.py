"""GPU: round-2 parity cases — container header accept / reject on the device parse, the 15-bit
code-length clamp in huffman_kernel, the stream decoder's header-first / surplus / multi-member / non-blocking
behaviour, handles of one context on two threads, shard decode of tiny shards, cfg4-shaped bit offsets beyond 2^32,
cfg5 and cfg2 at their full sizes."""
import io
import threading
import zlib as pyzlib

import numpy as np
import pytest

from golden import kat

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import os, sys
    import __graft_entry__ as g
    g.build()
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import synth
    import libflate_amd
    from libflate_amd import _ffi
    return libflate_amd, libflate_amd.Context(0), _ffi, synth


def same_verdict(got, want):
    # (status, bytes produced, consumed, message): the message is compared up to the first ':' (the reference's own
    # tests match prefixes only)
    return got[0] == want[0] and got[1] == want[1] and got[3].split(":")[0] == want[3].split(":")[0]


# ------------------------------------------------------------------ D-7: container headers on the device
def test_gzip_header_options_and_rejects(env, oracle):
    lfx, ctx, ffi, synth = env
    data = kat.test_i()[:20000]
    kw = dict(mtime=77, filename=b"x.txt", comment=b"hi there", hcrc=1, is_text=1, os=11,
              extra=bytes([0, 0x42, 3, 0]) + b"abc" + bytes([7, 7, 0, 0]))
    full = oracle.encode(oracle.GZIP, data, 8192, **kw)
    variants = {"all": full}
    for drop in ("filename", "comment", "hcrc", "extra"):
        k2 = dict(kw); k2.pop(drop)
        variants["no_" + drop] = oracle.encode(oracle.GZIP, data, 8192, **k2)
    for name, s in variants.items():
        got, want = ctx.decode_host(ffi.GZIP, s), oracle.decode(oracle.GZIP, s)
        assert want[0] == 0 and got[:3] == (0, data, len(s)), name            # gzip.rs:390-446 accepts
    hdr_len = full.index(oracle.encode(oracle.DEFLATE, data, 8192)[:8])
    bad = []
    for pos in range(hdr_len):                       # every header byte damaged once (ID, CM, FLG, fields, HCRC)
        b = bytearray(full); b[pos] ^= 0x10; bad.append(bytes(b))
    for cut in range(hdr_len + 2):                   # and every truncation inside / right behind the header
        bad.append(full[:cut])
    for s in bad:
        got, want = ctx.decode_host(ffi.GZIP, s), oracle.decode(oracle.GZIP, s)
        assert same_verdict(got, want), (len(s), got[0], want[0], got[3], want[3])
    # HCRC mismatch names both values (gzip.rs:437-441)
    b = bytearray(full); b[4] ^= 1
    rc, _, _, msg = ctx.decode_host(ffi.GZIP, bytes(b))
    assert rc == ffi.E_INVALID_DATA and msg.startswith("CRC16 of GZIP header mismatched")


def test_zlib_header_rejects(env, oracle):
    lfx, ctx, ffi, synth = env
    body = oracle.encode(oracle.DEFLATE, b"Hello World!") + (pyzlib.adler32(b"Hello World!")).to_bytes(4, "big")
    heads = [bytes([cmf, flg]) for cmf in (0x78, 0x68, 0x08, 0x88, 0x79, 0x77, 0xF8) for flg in (0x9C, 0x01, 0xDA, 0xBB, 0x20, 0x3F)]
    heads += [bytes([0x78, 0x20 | f]) for f in range(32)]          # FDICT with every FCHECK
    for h in heads:
        for s in (h + body, h + b"\x01\x02\x03\x04" + body, h, h[:1]):
            got, want = ctx.decode_host(ffi.ZLIB, s), oracle.decode(oracle.ZLIB, s)
            assert same_verdict(got, want), (h.hex(), len(s), got[0], want[0], got[3], want[3])


# ------------------------------------------------------------------ a-8: the 15-bit clamp on the device
def fib_skewed(n_sym=30, seed=5):
    f = [1, 1]
    while len(f) < n_sym:
        f.append(f[-1] + f[-2])
    data = np.concatenate([np.full(c, 33 + i, dtype=np.uint8) for i, c in enumerate(f)])
    np.random.default_rng(seed).shuffle(data)
    return data.tobytes()


def test_length_limit_clamp_runs_on_device(env, oracle):
    lfx, ctx, ffi, synth = env
    data = fib_skewed()                                  # 2.1 MB, byte frequencies 1,1,2,3,5,... (unconstrained depth 29)
    assert len(data) > (1 << 20)
    # literal-only parse (NoCompressionLz77Encoder, lib.rs:111-145): the block histogram IS the byte histogram → the
    # unconstrained depth exceeds 15 in every block; the default-LZ77 runs of the same data ride along
    freqs = np.bincount(np.frombuffer(data[:1 << 20], np.uint8), minlength=286); freqs[256] = 1
    assert max(oracle.huff_widths(list(freqs), 60)) > 15          # the limit is active (huffman.rs:202-209)
    for kw in (dict(lz77_kind=1), dict(lz77_kind=1, block_size=4 << 20), dict(), dict(block_size=4 << 20)):
        for fmt in (ffi.DEFLATE, ffi.ZLIB):
            got = ctx.encode_host(fmt, data, ffi.make_opts(**kw), ffi.make_schedule(8192))
            assert got == oracle.encode(fmt, data, write_size=8192, **kw), kw
            assert pyzlib.decompress(got, -15 if fmt == ffi.DEFLATE else 15) == data


# ------------------------------------------------------------------ a-2: single-chunk inputs whose walks do not merge
def test_single_chunk_parse_on_non_merging_data(env, oracle):
    """One write_all of several MiB = ONE LZ77 chunk with thousands of parse segments (the 1024-lane segment fold).  Runs of
    one byte and short periods make maximal-length matches whose greedy walks started at different phases never merge,
    so segments are re-walked from their true entry, some of them twice (the serial repair path inside the fold)."""
    lfx, ctx, ffi, synth = env
    rng = np.random.default_rng(12)
    text = synth.text(2 << 20, seed=0x5EED0031).tobytes()
    unit = rng.integers(0, 256, 257, dtype=np.uint8).tobytes()
    data = (b"\0" * (3 << 20) + text[:1 << 20] + unit * 6000 + bytes([7]) * 700001 + text[1 << 20:] +
            (b"ab" * 300000) + rng.integers(0, 3, 1 << 20, dtype=np.uint8).tobytes())
    for fmt, kw in ((ffi.GZIP, dict(mtime=0)), (ffi.DEFLATE, dict(max_length=100)), (ffi.ZLIB, dict(window_size=4096))):
        got = ctx.encode_host(fmt, data, ffi.make_opts(**kw), ffi.make_schedule(0))
        want = oracle.encode(fmt, data, write_size=0, **kw)
        assert got == want, (fmt, kw, len(got), len(want))
        rc, out = ctx.decode_host(fmt, got)[:2]
        assert rc == 0 and out == data


# ------------------------------------------------------------------ b-3: stream decoder
class ChunkReader:
    """hands out at most `step` bytes per read() — and counts what it was asked for"""
    def __init__(self, data, step):
        self.data, self.step, self.pos, self.calls = data, step, 0, 0

    def read(self, n):
        self.calls += 1
        k = min(n, self.step, len(self.data) - self.pos)
        b = self.data[self.pos:self.pos + k]
        self.pos += k
        return b


class WouldBlockReader(ChunkReader):
    """src/util.rs:7-44: alternates WouldBlock and a piece of data"""
    def read(self, n):
        if self.calls % 2 == 0:
            self.calls += 1
            raise BlockingIOError()
        return ChunkReader.read(self, n)


def test_stream_decoder_high_ratio_and_sizes(env, oracle):
    lfx, ctx, ffi, synth = env
    zeros = bytes(9 << 20)                               # ~1000:1: the first capacity guess is too small
    for mod, fmt in ((lfx.gzip, oracle.GZIP), (lfx.zlib, oracle.ZLIB), (lfx.deflate, oracle.DEFLATE)):
        s = oracle.encode(fmt, zeros, 8192)
        assert len(s) < len(zeros) // 500
        assert mod.Decoder.new(s).read_to_end() == zeros
        assert mod.Decoder.new(ChunkReader(s, 1000)).read_to_end() == zeros
    text = synth.text(3 << 20).tobytes()
    s = oracle.encode(oracle.GZIP, text, 8192)
    for step in (1 << 30, 65536, 4096, 100000):
        r = ChunkReader(s, step)
        d = lfx.gzip.Decoder.new(r)
        assert r.pos <= max(65536, step) or step > len(s)          # header first: not the whole stream
        out = b""
        while True:
            b = d.read(50000)
            if not b:
                break
            out += b
        assert out == text and d.consumed() == len(s) and d.surplus() == b""


def test_stream_decoder_header_and_surplus(env, oracle):
    lfx, ctx, ffi, synth = env
    a, b = synth.text(200000).tobytes(), kat.test_i()
    kw = dict(mtime=1234567, filename=b"a.txt", comment=b"first", hcrc=1, is_text=1, os=7, extra=bytes([65, 66, 2, 0, 9, 8]))
    sa, sb = oracle.encode(oracle.GZIP, a, 8192, **kw), oracle.encode(oracle.GZIP, b, 0, mtime=5)
    d = lfx.gzip.Decoder.new(sa + sb + b"trailing")
    h = d.header()                                                  # gzip.rs:959 — available before any read()
    assert (h["modification_time"], h["os"], h["is_text"], h["is_verified"]) == (1234567, 7, True, True)
    assert h["filename"] == b"a.txt" and h["comment"] == b"first" and h["extra_field"] == bytes([65, 66, 2, 0, 9, 8])
    assert d.read_to_end() == a
    assert d.consumed() == len(sa) and d.surplus() == sb + b"trailing"       # gzip.rs:1216-1226
    m = lfx.gzip.MultiDecoder.new(ChunkReader(sa + sb, 70000))
    assert m.header()["filename"] == b"a.txt"
    assert m.read_to_end() == a + b and m.consumed() == len(sa) + len(sb)
    assert m.header()["modification_time"] == 5 and m.header()["filename"] is None   # header of the LAST member (gzip.rs:1106)
    z = lfx.zlib.Decoder.new(oracle.encode(oracle.ZLIB, a, 8192, window_size=4096))
    assert z.header() == {"window_size": 4096, "compression_level": 2}            # zlib.rs:335
    assert z.read_to_end() == a
    # constructor failures: header errors surface in new() (gzip.rs:941-944, zlib.rs:312-320) ...
    for mod, junk in ((lfx.gzip, b"\x1f\x8c" + sa[2:]), (lfx.gzip, sa[:7]), (lfx.zlib, b"\x78\x9d1234"), (lfx.gzip, b"")):
        with pytest.raises(lfx.gzip.StreamError):
            mod.Decoder.new(junk)
    # ... body errors in read(), after the bytes of the completed blocks
    cut = sa[:len(sa) // 2]
    d = lfx.gzip.Decoder.new(cut)
    want = oracle.decode(oracle.GZIP, cut)
    with pytest.raises(lfx.gzip.StreamError) as ei:
        d.read_to_end()
    assert ei.value.kind == "UnexpectedEof" and ei.value.partial + d.unread_decoded_data() == want[1]
    assert d.read(10) == b""                                         # the error is reported once


def test_non_blocking_decoders(env, oracle):
    lfx, ctx, ffi, synth = env
    text = synth.text(400000).tobytes()
    for mod, fmt in ((lfx.non_blocking.gzip, oracle.GZIP), (lfx.non_blocking.zlib, oracle.ZLIB),
                     (lfx.non_blocking.deflate, oracle.DEFLATE)):
        s = oracle.encode(fmt, text, 8192)
        r = WouldBlockReader(s + b"rest", 30000)
        d = mod.Decoder.new(r)                       # reads nothing yet (non_blocking/gzip.rs:64-88)
        assert r.calls == 0
        out, blocks = b"", 0
        while True:                                  # src/util.rs:46-66 nb_read_to_end
            try:
                b = d.read(100000)
            except BlockingIOError:
                blocks += 1
                continue
            if not b:
                break
            out += b
        assert out == text and blocks > 0
        assert d.consumed() == len(s) and d.surplus().endswith(b"rest")
    # the header getter of a non-blocking gzip decoder may itself have to wait
    s = oracle.encode(oracle.GZIP, text, 8192, filename=b"n" * 3000)
    d = lfx.non_blocking.gzip.Decoder.new(WouldBlockReader(s, 1000))
    tries = 0
    while True:
        try:
            h = d.header()
            break
        except BlockingIOError:
            tries += 1
    assert tries >= 2 and h["filename"] == b"n" * 3000
    # reject vectors keep their verdict through the non-blocking path (non_blocking/deflate/decode.rs:283-299)
    for v in (kat.TOO_LONG_BACKREF, kat.ISSUE64):
        d = lfx.non_blocking.deflate.Decoder.new(WouldBlockReader(v, 7))
        with pytest.raises(lfx.gzip.StreamError) as ei:
            while True:
                try:
                    if not d.read(1000):
                        break
                except BlockingIOError:
                    pass
        assert ei.value.kind == "InvalidData"


def test_two_handles_two_threads(env, oracle):
    """SURVEY §8b threading: a handle is single-threaded, distinct handles are independent — also on ONE context."""
    lfx, ctx, ffi, synth = env
    datas = [synth.text(3 << 20, seed=synth.SEED_BASE + 40 + i).tobytes() for i in range(4)]
    wants = [oracle.encode(oracle.GZIP if i % 2 else oracle.ZLIB, d, 8192) for i, d in enumerate(datas)]
    results, errors = [None] * 4, []

    def work(i):
        try:
            for _ in range(3):
                mod = lfx.gzip if i % 2 else lfx.zlib
                sink = io.BytesIO()
                e = mod.Encoder.new(sink, context=ctx)
                for off in range(0, len(datas[i]), 8192):
                    e.write(datas[i][off:off + 8192])
                e.finish()
                assert sink.getvalue() == wants[i]
                assert mod.Decoder.new(wants[i], context=ctx).read_to_end() == datas[i]
            results[i] = True
        except Exception as ex:  # noqa: BLE001
            errors.append((i, repr(ex)))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors and all(results), errors


def test_stream_encoder_batches(env, oracle):
    """The stream encoder hands closed blocks to the sink as they accumulate (8 MiB batches — the reference emits per
    block, encode.rs:277-286; stored blocks every 1024 blocks) instead of holding everything until finish(): what it
    buffers stays within two batches (one of them in flight on the GPU) plus the open block."""
    lfx, ctx, ffi, synth = env
    data = synth.text(150 << 20).tobytes()
    sink = io.BytesIO()
    e = lfx.gzip.Encoder.new(sink, context=ctx)
    seen = []
    for off in range(0, len(data), 1 << 20):
        e.write(data[off:off + (1 << 20)])
        seen.append(sink.tell())
    e.finish()
    # only the header until 8 MiB of closed blocks wait; that batch is STARTED by the write that closes it (round 6: one batch
    # in flight on the GPU while the caller copies the next one in) and reaches the sink when the next batch is started
    assert seen[6] == 10 and seen[8] == 10 and seen[15] > (2 << 20) and seen[-1] > (40 << 20)
    # ... and from then on the sink is never more than two batches and the open block behind the writer (C / N ≈ 0.48)
    for i in range(18, len(seen)):
        assert seen[i] > ((i + 1 - 18) << 20) * 0.40, (i, seen[i])
    assert pyzlib.decompress(sink.getvalue(), 31) == data
    assert sink.getvalue()[:1 << 20] == oracle.encode(oracle.GZIP, data[:8 << 20], 1 << 20)[:1 << 20]
    raw = synth.text(80 << 20).tobytes()
    sink = io.BytesIO()
    e = lfx.deflate.Encoder.with_options(sink, lfx.deflate.EncodeOptions().no_compression())
    marks = []
    for off in range(0, len(raw), 1 << 20):
        e.write(raw[off:off + (1 << 20)])
        marks.append(sink.tell())
    e.finish()
    assert marks[70] > 60 << 20                               # stored blocks leave as they are closed
    assert pyzlib.decompress(sink.getvalue(), -15) == raw


# ------------------------------------------------------------------ e: shards
def shard_roundtrip(env, oracle, world, n, fake_lead_bits=0):
    import ctypes as C
    import torch
    lfx, ctx, ffi, synth = env
    from libflate_amd import sharded
    datas = [synth.text(n, seed=synth.SEED_BASE + 70 + r) for r in range(world)]
    opts, sched = ffi.make_opts(mtime=0), ffi.make_schedule(8192)
    L = ffi.lib()
    d_ins = [torch.from_numpy(d).cuda() for d in datas]
    ctxs = [lfx.Context(0) for _ in range(world)]
    infos = []
    for r in range(world):
        info = ffi.ShardInfo()
        rc = L.lfx_encode_shard_prepare(ctxs[r].handle, ffi.GZIP, C.byref(opts), C.byref(sched), d_ins[r].data_ptr(), n,
                                        int(r == 0), int(r == world - 1), C.byref(info))
        assert rc == 0, ctxs[r].last_error()
        infos.append((info.total_bits, info.n_bytes, info.crc32, info.adler32))
    hdr_len = L.lfx_container_header_len(ffi.GZIP, C.byref(opts))
    start_bits, check, total_n = sharded.layout(infos, hdr_len, ffi.GZIP)
    parts, outs = [], []
    for r in range(world):
        cap = n + n // 4 + 65536
        d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        m = C.c_uint64(0)
        # ranks behind the first see a start bit far beyond 2^32 (as if the shards in front were 4+ GiB of
        # stream); a multiple of 8 keeps the bit phase, so the bytes must not change
        sb = start_bits[r] + (fake_lead_bits if r else 0)
        rc = L.lfx_encode_shard_emit(ctxs[r].handle, sb, check, total_n, d_out.data_ptr(), cap, C.byref(m))
        assert rc == 0, ctxs[r].last_error()
        parts.append(d_out[:m.value].cpu().numpy().tobytes())
        outs.append((d_out, m.value, sb))
    member = sharded.assemble(parts, start_bits)
    whole = b"".join(d.tobytes() for d in datas)
    want = oracle.encode(oracle.GZIP, whole, write_size=8192)
    assert member == want
    # the same concatenation on the device (what the writer rank does with the shards it received over xGMI)
    real_bits = [start_bits[r] for r in range(world)]
    cap = len(want) + 64
    d_member = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    for r in range(world):
        rc = L.lfx_shard_place_device(ctxs[0].handle, d_member.data_ptr(), cap, outs[r][0].data_ptr(), outs[r][1],
                                      real_bits[r], int(r == 0))
        assert rc == 0, ctxs[0].last_error()
    assert d_member[:sharded.member_bytes(real_bits, [o[1] for o in outs])].cpu().numpy().tobytes() == want
    for r in range(world):
        d_dec = torch.zeros(n, dtype=torch.uint8, device="cuda")
        ol = C.c_uint64(0)
        sb = outs[r][2] if r == 0 else outs[r][2] & 7
        rc = L.lfx_decode_shard_device(ctxs[r].handle, outs[r][0].data_ptr(), outs[r][1], sb, infos[r][0],
                                       int(r == world - 1), d_dec.data_ptr(), n, C.byref(ol))
        assert rc == 0 and ol.value == n, (r, ctxs[r].last_error())
        assert torch.equal(d_dec, d_ins[r])


def test_cfg4_shaped_eight_ranks_large_bit_offsets(env, oracle):
    shard_roundtrip(env, oracle, 8, 4 << 20, fake_lead_bits=(1 << 35) + (1 << 33))


def test_tiny_shards_decode_alone(env, oracle):
    # shards whose compressed size is far below the block finder's threshold: the exact walk, ended at the shard's last bit
    shard_roundtrip(env, oracle, 3, 1 << 20)


def test_total_bits_beyond_2_32(env):
    """cfg4 arithmetic at scale: ONE shard whose own DEFLATE stream is longer than 2^32 bits (1 GiB of random
    bytes: C/N ~ 1.0), checked by python zlib (independent inflater + CRC) and the GPU round trip."""
    import torch
    lfx, ctx, ffi, synth = env
    n = 1 << 30
    g = torch.Generator(device="cuda"); g.manual_seed(1234)
    d_in = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda", generator=g)
    sched, opts = ffi.make_schedule(8192), ffi.make_opts()
    bound = ffi.lib().lfx_encode_bound(n, None, None) & ~3
    d_out = torch.empty(bound, dtype=torch.uint8, device="cuda")
    m = ctx.encode_device(ffi.GZIP, d_in.data_ptr(), n, d_out.data_ptr(), bound, opts, sched)
    assert m * 8 > (1 << 32)
    comp = d_out[:m].cpu().numpy().tobytes()
    plain = pyzlib.decompress(comp, 31)
    host = d_in.cpu().numpy().tobytes()
    assert len(plain) == n and plain == host
    del plain, host
    assert int.from_bytes(comp[-4:], "little") == n & 0xFFFFFFFF
    d_dec = torch.zeros(n, dtype=torch.uint8, device="cuda")
    rc, ol, used, msg = ctx.decode_device(ffi.GZIP, d_out.data_ptr(), m, d_dec.data_ptr(), n)
    assert (rc, ol, used) == (0, n, m), msg
    assert torch.equal(d_dec, d_in)


# ------------------------------------------------------------------ full-size configurations
def test_cfg5_full_size_1gib_lowent(env):
    import torch
    lfx, ctx, ffi, synth = env
    n = 1 << 30
    data = synth.lowent(n)
    d_in = torch.from_numpy(data).cuda()
    sched, opts = ffi.make_schedule(8192), ffi.make_opts()
    bound = ffi.lib().lfx_encode_bound(n, None, None) & ~3
    d_out = torch.empty(bound, dtype=torch.uint8, device="cuda")
    m = ctx.encode_device(ffi.ZLIB, d_in.data_ptr(), n, d_out.data_ptr(), bound, opts, sched)
    comp = d_out[:m].cpu().numpy().tobytes()
    assert m < n // 10
    plain = pyzlib.decompress(comp)                               # independent inflater, verifies Adler-32 itself
    assert len(plain) == n and pyzlib.adler32(plain) == int.from_bytes(comp[-4:], "big")
    assert pyzlib.crc32(plain) == pyzlib.crc32(data.tobytes())
    del plain
    d_dec = torch.zeros(n, dtype=torch.uint8, device="cuda")
    rc, ol, used, msg = ctx.decode_device(ffi.ZLIB, d_out.data_ptr(), m, d_dec.data_ptr(), n)
    assert (rc, ol, used) == (0, n, m), msg
    assert torch.equal(d_dec, d_in)


@pytest.mark.parametrize("write_size", [8192, 0])
def test_cfg2_256mib_bit_exact_vs_oracle(env, oracle, write_size):
    """the benchmarked configuration, byte for byte (S8K) — and its S1 schedule (one write_all)"""
    import torch
    lfx, ctx, ffi, synth = env
    n = 256 << 20
    data = synth.text(n)
    d_in = torch.from_numpy(data).cuda()
    sched, opts = ffi.make_schedule(write_size), ffi.make_opts()
    bound = ffi.lib().lfx_encode_bound(n, C_byref(opts), C_byref(sched)) & ~3
    d_out = torch.empty(bound, dtype=torch.uint8, device="cuda")
    m = ctx.encode_device(ffi.GZIP, d_in.data_ptr(), n, d_out.data_ptr(), bound, opts, sched)
    got = d_out[:m].cpu().numpy().tobytes()
    assert got == oracle.encode(oracle.GZIP, data.tobytes(), write_size=write_size)
    d_dec = torch.zeros(n, dtype=torch.uint8, device="cuda")
    rc, ol, used, msg = ctx.decode_device(ffi.GZIP, d_out.data_ptr(), m, d_dec.data_ptr(), n)
    assert (rc, ol, used) == (0, n, m), msg
    assert torch.equal(d_dec, d_in)


def C_byref(x):
    import ctypes
    return ctypes.byref(x)


def test_shim_abi_program(env):
    """tests/c/shim_abi.c: every extern "C" function the Rust shim crate declares, driven from C on the reference's
    known-answer vectors (the crate itself cannot be compiled in this image)."""
    import subprocess
    from test_abi import _shim_binary, _shim_env
    out = subprocess.run([_shim_binary()], env=_shim_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "shim abi ok" in out.stdout, (out.stdout, out.stderr)
    # round 6: lfx_comm_rccl EXECUTED on a one-rank RCCL communicator (all-gather, grouped self send / recv, start / wait,
    # both drivers) — torch ships librccl, so "skipped" would mean the loader path is broken
    assert "rccl binding ok" in out.stdout, (out.stdout, out.stderr)
