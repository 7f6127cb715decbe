"""GPU: round-4 parity cases — a caller-supplied Lz77Encode through the GPU Huffman / pack stages (lfx_encoder_write_codes,
EncodeOptions::with_lz77(E) src/deflate/encode.rs:59-65), the N-GPU decode of members whose blocks read across rank
boundaries (window hand-over), stream decoding of long fixed-Huffman members (blocks cut by a window's end)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def env():
    import __graft_entry__ as g
    g.build()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth
    import libflate_amd
    from libflate_amd import _ffi
    return libflate_amd, libflate_amd.Context(0), _ffi, synth


# ------------------------------------------------------------------ b-1 / a-7..a-14: arbitrary E: Lz77Encode
class OddChunksLiteral:
    """A non-default Lz77Encode: buffers like DefaultLz77Encoder (default.rs:60-68, flush at >= window * 8 bytes), but
    every SECOND flush unit is emitted as literals only, the others with the oracle's greedy parse at max_length 100."""

    def __init__(self, oracle, level, window=4096):
        self.o, self.level, self.window = oracle, level, window
        self.buf = bytearray()
        self.k = 0

    def encode(self, buf, sink):
        self.buf += buf
        if len(self.buf) >= self.window * 8:
            self.flush(sink)

    def flush(self, sink):
        data = bytes(self.buf)
        self.buf.clear()
        self.k += 1
        if self.k % 2 == 0:
            sink.extend(("Literal", b) for b in data)
        else:
            for w in self.o.lz77_chunk(data, self.window, 100):
                w = int(w)
                sink.append(("Literal", w >> 16) if (w & 0xFFFF) == 0 else ("Pointer", w >> 16, w & 0xFFFF))

    def compression_level(self):
        return self.level

    def window_size(self):
        return self.window


def test_with_lz77_foreign_encoder_vs_oracle(env, oracle):
    """EncodeOptions::with_lz77(E) for an E that is not ours (encode.rs:59-65): E runs on the caller's side, the GPU
    Huffman-codes whatever it emits (CompressBuf::{append,flush}, encode.rs:405-425).  Same E through the oracle's generic
    parameter: byte-identical streams, all three containers, dynamic and fixed codes, flushes, both zlib flush modes; the
    container headers carry E's window size and compression level (zlib.rs:212-220, gzip.rs:684)."""
    import io
    import zlib
    lfx, ctx, ffi, synth = env
    data = synth.text(1 << 20).tobytes() + synth.lowent(150000).tobytes()
    rng = np.random.default_rng(5)
    cuts = sorted(set(int(x) for x in rng.integers(0, len(data), 40)) | {0, len(data)})
    for fmt, mod, ofmt in ((ffi.DEFLATE, lfx.deflate, oracle.DEFLATE), (ffi.ZLIB, lfx.zlib, oracle.ZLIB), (ffi.GZIP, lfx.gzip, oracle.GZIP)):
        for variant in ("dynamic", "fixed", "small-blocks", "sync"):
            if variant == "sync" and fmt != ffi.ZLIB:
                continue
            level = {"dynamic": 3, "fixed": 1, "small-blocks": 0, "sync": 2}[variant]
            okw = dict(oracle.custom_lz77(OddChunksLiteral(oracle, level)))
            opts = mod.EncodeOptions().with_lz77(OddChunksLiteral(oracle, level))
            if variant == "fixed":
                okw["dynamic_huffman"] = 0
                opts = opts.fixed_huffman_codes()
            if variant == "small-blocks":
                okw["block_size"] = 50000
                opts = opts.block_size(50000)
            if variant == "sync":
                okw["zlib_sync_flush"] = 1
                opts = opts.flush_mode(lfx.zlib.FlushMode.SYNC)
            if fmt == ffi.GZIP:
                okw["mtime"] = 0
            ref = oracle.Encoder(ofmt, **okw)
            sink = io.BytesIO()
            enc = mod.Encoder.with_options(sink, opts)
            for i, (a, b) in enumerate(zip(cuts, cuts[1:])):
                ref.write(data[a:b])
                enc.write(data[a:b])
                if i % 7 == 3:
                    ref.flush()
                    enc.flush()
            want = ref.finish()
            enc.finish()
            got = sink.getvalue()
            assert got == want, (fmt, variant, len(got), len(want))
            wb = -15 if fmt == ffi.DEFLATE else 15 if fmt == ffi.ZLIB else 31
            assert zlib.decompress(got, wb) == data


def test_write_codes_rejects(env):
    """Code words outside Code's domain (lib.rs:27-42), bytes after codes, an unclosed block at finish: LFX_E_ARG."""
    import ctypes as C
    lfx, ctx, ffi, synth = env
    L = ffi.lib()
    got = []
    wcb = ffi.WRITE_CB(lambda u, p, n: (got.append(C.string_at(p, n)), n)[1])
    fcb = ffi.FLUSH_CB(lambda u: 0)
    st = C.c_int(0)
    opts = ffi.make_opts()
    for bad in ((300 << 16), (2 << 16) | 5, (259 << 16) | 1, (3 << 16) | 32769):
        e = L.lfx_encoder_new(ctx.handle, ffi.DEFLATE, C.byref(opts), wcb, fcb, None, C.byref(st))
        w = (C.c_uint32 * 1)(bad)
        assert L.lfx_encoder_write_codes(e, w, 1, b"x", 1, 0) == ffi.E_ARG
        L.lfx_encoder_free(e)
    e = L.lfx_encoder_new(ctx.handle, ffi.DEFLATE, C.byref(opts), wcb, fcb, None, C.byref(st))
    w = (C.c_uint32 * 2)(ord("a") << 16, ord("b") << 16)
    assert L.lfx_encoder_write_codes(e, w, 2, b"ab", 2, 0) == 0
    assert L.lfx_encoder_write(e, b"zz", 2) == -ffi.E_ARG          # bytes and codes do not mix
    assert L.lfx_encoder_finish(e) == ffi.E_ARG                    # the final block was never closed
    assert L.lfx_encoder_write_codes(e, w, 0, None, 0, 2) == 0
    assert L.lfx_encoder_finish(e) == 0
    import zlib
    assert zlib.decompress(b"".join(got), -15) == b"ab"
    L.lfx_encoder_free(e)


# ------------------------------------------------------------------ e: N-GPU decode, window hand-over
def test_foreign_member_on_virtual_ranks(env, oracle):
    """Members of another encoder (python zlib, levels 1 / 6 / 9): their blocks read up to 32 KiB of earlier output, across
    block and RANK boundaries (the reference decodes any valid stream: decode.rs:112-164, lib.rs:164-194).  Every rank
    materialises its slice as symbols, the ranks exchange one 64 KiB index map each, every rank composes the window in
    front of its slice.  Also: ranks that own no block at all (more ranks than blocks), a member
    that mixes reference-made and foreign halves is not needed — the per-rank state decides the path."""
    import gzip as pygzip
    import zlib
    import torch
    from test_gpu_round3 import _virtual_rank_decode
    lfx, ctx, ffi, synth = env
    text = synth.text(24 << 20)
    cases = [("text-6", text.tobytes(), 6, (2, 4, 8)), ("text-1", text[:8 << 20].tobytes(), 1, (3,)),
             ("text-9", text[:6 << 20].tobytes(), 9, (4,)), ("lowent-6", synth.lowent(12 << 20).tobytes(), 6, (4,)),
             ("small-6", text[:700000].tobytes(), 6, (8,))]
    for name, plain, level, worlds in cases:
        member = pygzip.compress(plain, level, mtime=0)
        want = torch.frombuffer(bytearray(plain), dtype=torch.uint8).cuda()
        for world in worlds:
            out, (crc, ad), owned, nch, ncand = _virtual_rank_decode(lfx, ffi, member, 10, world, len(plain))
            assert torch.equal(out, want), (name, world)
            assert crc == zlib.crc32(plain) == int.from_bytes(member[-8:-4], "little"), (name, world)
            assert sum(owned) == len(plain)


# ------------------------------------------------------------------ b-3 / D-6: windows that cut fixed-Huffman blocks
def test_stream_decoder_long_fixed_huffman_member(env):
    """a window's end cuts a block in half; the scan's last lane reads a few symbols past the input and a
    garbage symbol that reads as EndOfBlock (7 bits in a fixed-Huffman block: about 6 % per cut) must not make the cut
    block look complete.  A 160 MiB member of 1 MiB fixed-Huffman blocks (fixed_huffman_codes(), encode.rs:107-110)
    streamed through 16 MiB windows with two reader step sizes; and one of 8 MiB blocks (every window cuts one)."""
    import ctypes as C
    import zlib
    import torch
    from test_gpu_round3 import _stream_decode
    lfx, ctx, ffi, synth = env
    n = 160 << 20
    data = synth.text(n)
    crc_want = zlib.crc32(data.tobytes())
    d_in = torch.from_numpy(data).cuda()
    for block_size in (1 << 20, 8 << 20):
        opts, sched = ffi.make_opts(mtime=0, dynamic_huffman=0, block_size=block_size), ffi.make_schedule(8192)
        bound = ffi.lib().lfx_encode_bound(n, C.byref(opts), C.byref(sched)) & ~3
        d_out = torch.empty(bound, dtype=torch.uint8, device="cuda")
        m = ctx.encode_device(ffi.GZIP, d_in.data_ptr(), n, d_out.data_ptr(), bound, opts, sched)
        stream = d_out[:m].cpu().numpy().tobytes()
        for step in ((1 << 20) + 4099, 3 << 20):
            crc, total, first_pos, peak, status, consumed = _stream_decode(ctx, ffi, ffi.GZIP, stream, step, 8 << 20)
            assert status == 0 and total == n and crc == crc_want and consumed == len(stream), (block_size, step, status, total)


# ------------------------------------------------------------------ b-4: batch encode
def test_encode_batch_vs_oracle(env, oracle):
    """lfx_encode_batch_device: many independent streams in one launch set.  Stream i must be what the reference makes of its
    bytes alone (Encoder::new + write_all + finish per stream: encode.rs:182-249, zlib.rs:577-681, gzip.rs:804-908) — ragged
    sizes including empty and tiny ones, all three containers, both schedules, fixed codes and stored blocks; a stream
    whose capacity is too small voids the call with LFX_E_NOSPACE and status[] names it."""
    import ctypes as C
    import torch
    lfx, ctx, ffi, synth = env
    L = ffi.lib()
    rng = np.random.default_rng(3)
    sizes = [0, 1, 2, 3, 4, 70000, 65536, 65535, 8192, 8193, 300000, 13312, 5, 262144, 262147, 40000] + [int(x) for x in rng.integers(1, 100000, 40)]
    text = synth.text(sum(sizes) + 16).tobytes()
    low = synth.lowent(300000).tobytes()
    bufs, at = [], 0
    for k, n in enumerate(sizes):
        bufs.append(low[:n] if k % 5 == 4 else text[at:at + n])
        at += n
    count = len(bufs)
    in_off = np.zeros(count, dtype=np.uint64)
    in_len = np.array([len(b) for b in bufs], dtype=np.uint64)
    pos = 0
    for i, b in enumerate(bufs):
        in_off[i] = pos
        pos += len(b) + (7 * i) % 5                    # (gaps and unaligned starts)
    host = np.zeros(pos + 8, dtype=np.uint8)
    for i, b in enumerate(bufs):
        host[int(in_off[i]):int(in_off[i]) + len(b)] = np.frombuffer(b, dtype=np.uint8)
    d_in = torch.from_numpy(host).cuda()
    cases = [(ffi.ZLIB, oracle.ZLIB, {}, {}, 0), (ffi.GZIP, oracle.GZIP, {"mtime": 7}, {"mtime": 7}, 8192), (ffi.DEFLATE, oracle.DEFLATE, {"dynamic_huffman": 0}, {"dynamic_huffman": 0}, 0),
             (ffi.ZLIB, oracle.ZLIB, {"no_compression": 1}, {"no_compression": 1}, 8192), (ffi.DEFLATE, oracle.DEFLATE, {"block_size": 30000}, {"block_size": 30000}, 1000)]
    for fmt, ofmt, kw, okw, ws in cases:
        opts, sched = ffi.make_opts(**kw), ffi.make_schedule(ws)
        out_cap = np.array([(L.lfx_encode_bound(len(b), C.byref(opts), C.byref(sched)) + 3) & ~3 for b in bufs], dtype=np.uint64)
        out_off = np.concatenate(([0], np.cumsum(out_cap)[:-1])).astype(np.uint64)
        d_out = torch.full((int(out_cap.sum()),), 0xAA, dtype=torch.uint8, device="cuda")
        out_len = np.zeros(count, dtype=np.uint64)
        status = np.zeros(count, dtype=np.int32)
        rc = L.lfx_encode_batch_device(ctx.handle, fmt, C.byref(opts), C.byref(sched), count, d_in.data_ptr(), in_off.ctypes.data,
                                       in_len.ctypes.data, d_out.data_ptr(), out_off.ctypes.data, out_cap.ctypes.data, out_len.ctypes.data,
                                       status.ctypes.data)
        assert rc == 0 and not status.any(), (rc, ctx.last_error())
        got = d_out.cpu().numpy()
        for i, b in enumerate(bufs):
            want = oracle.encode(ofmt, b, write_size=ws, **okw)
            assert got[int(out_off[i]):int(out_off[i]) + int(out_len[i])].tobytes() == want, (fmt, kw, i, len(b))
        # one stream too small: nothing is written, the call and the stream say NOSPACE
        small = out_cap.copy()
        small[5] = 64
        rc = L.lfx_encode_batch_device(ctx.handle, fmt, C.byref(opts), C.byref(sched), count, d_in.data_ptr(), in_off.ctypes.data,
                                       in_len.ctypes.data, d_out.data_ptr(), out_off.ctypes.data, small.ctypes.data, out_len.ctypes.data,
                                       status.ctypes.data)
        assert rc == ffi.E_NOSPACE and status[5] == ffi.E_NOSPACE and not out_len.any()
        assert status[4] == 0 and status[6] == 0        # (only the stream that was too small is named)


# ------------------------------------------------------------------ D-6: mid-size members (a few ordinary blocks)
def test_mid_size_members_scanned_in_pieces(env, oracle):
    """A member of 9 … 128 ordinary blocks (8 … 100 MiB at the default block size) is scanned in pieces over every candidate
    range (lfx_decode.cpp, round 4) and materialised through the marker path: same bytes as the input, same verdict and
    delivered prefix as the oracle's on a truncated and on a corrupted member (decode.rs:112-164), and the
    one-workgroup-per-block path (LFX_NO_PIECES) agrees."""
    import zlib
    import torch
    lfx, ctx, ffi, synth = env
    for name, data in (("text-12m", synth.text(12 << 20).tobytes()), ("lowent-20m", synth.lowent(20 << 20).tobytes()),
                       ("text-40m", synth.text(40 << 20).tobytes())):
        stream = ctx.encode_host(ffi.GZIP, data, ffi.make_opts(mtime=0), ffi.make_schedule(8192))
        rc, out = ctx.decode_host(ffi.GZIP, stream)[:2]
        assert rc == 0 and out == data, (name, rc, len(out))
    data = synth.text(9 << 20).tobytes()
    member = oracle.encode(oracle.ZLIB, data, write_size=8192)
    assert zlib.decompress(member) == data
    rc, out = ctx.decode_host(ffi.ZLIB, member)[:2]
    assert rc == 0 and out == data
    cut = member[:len(member) * 2 // 3]
    bad = bytearray(member); bad[len(member) // 2] ^= 0x10; bad = bytes(bad)
    for name, s in (("truncated", cut), ("corrupted", bad)):
        want = oracle.decode(oracle.ZLIB, s)
        got = ctx.decode_host(ffi.ZLIB, s)
        assert got[0] == {1: ffi.E_INVALID_DATA, 2: ffi.E_UNEXPECTED_EOF}[want[0]], (name, got[0], want[0], want[3])
        assert got[1] == want[1], (name, len(got[1]), len(want[1]))


def test_pieces_survive_a_false_candidate(env):
    """About one bit offset per 30 MB of stream passes the block finder without being a block start.  The 48 MiB text of
    the synthetic corpus holds one (measured, round 4): the candidate range it cuts in two has no EndOfBlock, and the pieces
    behind it were scanned with tables read from data — the range is scanned again as one and the member still takes the piece
    path (phases: `pieces`, no `blk_scan`).  Sizes on both sides of it for comparison; all of them byte-exact."""
    import torch
    lfx, ctx, ffi, synth = env
    for mib in (40, 48, 56):
        data = synth.text(mib << 20)
        d_in = torch.from_numpy(data).cuda()
        opts, sched = ffi.make_opts(mtime=0), ffi.make_schedule(8192)
        import ctypes as C
        bound = ffi.lib().lfx_encode_bound(data.size, C.byref(opts), C.byref(sched))
        d_out = torch.empty(bound, dtype=torch.uint8, device="cuda")
        m = ctx.encode_device(ffi.GZIP, d_in.data_ptr(), data.size, d_out.data_ptr(), bound, opts, sched)
        d_dec = torch.zeros(data.size, dtype=torch.uint8, device="cuda")
        ctx.enable_timing(True)
        rc, ol, used, msg = ctx.decode_device(ffi.GZIP, d_out.data_ptr(), m, d_dec.data_ptr(), data.size)
        phases = [k for k, _ in (ctx.last_timing() or {"phases": []})["phases"]]
        ctx.enable_timing(False)
        assert rc == 0 and ol == data.size and used == m, (mib, rc, msg)
        assert torch.equal(d_dec, d_in), mib
        assert "pieces" in phases and "blk_scan" not in phases, (mib, phases)


def test_with_lz77_foreign_encoder_many_codes(env, oracle):
    """More than ENC_BATCH_CODES (2 Mi) code words between two flushes: closed blocks leave the encoder in batches while later
    ones are still being collected (the carry of a partial last byte, the running checksum across batches)."""
    import io
    lfx, ctx, ffi, synth = env

    class EveryByteALiteral:
        def encode(self, buf, sink):
            sink.extend(("Literal", b) for b in buf)

        def flush(self, sink):
            pass

        def compression_level(self):
            return 0

        def window_size(self):
            return 32768

    data = synth.text(5 << 19).tobytes()                       # 2.5 MiB = 2.6 M codes, blocks of 300000 bytes
    want = oracle.encode(oracle.GZIP, data, write_size=70000, mtime=0, block_size=300000, **oracle.custom_lz77(EveryByteALiteral()))
    sink = io.BytesIO()
    enc = lfx.gzip.Encoder.with_options(sink, lfx.gzip.EncodeOptions().with_lz77(EveryByteALiteral()).block_size(300000))
    for a in range(0, len(data), 70000):
        enc.write(data[a:a + 70000])
    enc.finish()
    assert sink.getvalue() == want
    # ... and the same stream is what the device pipeline makes for NoCompressionLz77Encoder (lib.rs:111-145)
    assert want == oracle.encode(oracle.GZIP, data, write_size=70000, mtime=0, block_size=300000, lz77_kind=1)
