"""GPU: round-6 cases — the drop-in surface on HOST memory.

* cfg1 as SURVEY.md §8d words it: `gzip.compress(TEXT(1 MiB), mtime=0)` made by python, decoded through the STREAM ABI with
  8 KiB reads (lfx_decoder_read; the reference's `io::copy` out of a `gzip::Decoder`, src/gzip.rs:1018-1047,
  examples/flate.rs:96-97) — and by the oracle on the CPU; both must give the input back.
* lfx_encode_host / lfx_decode_host on pageable and on page-locked buffers (lfx_hostio.h): the bytes of the oracle at sizes
  around the staging thresholds.
* The io::copy protocol (8192-byte write() calls, examples/flate.rs:52) through the stream encoder with page-locked pending /
  output buffers, at several batch sizes: the oracle's bytes for the same write schedule."""
import ctypes as C
import gzip as pygzip
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_gpu_parity import ctx, enc, ffi, lfx, synth  # noqa: F401  (fixtures)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_cfg1_python_gzip_1mib_stream_api_8k_reads(ctx, ffi, oracle, synth):
    import stream_copy
    data = synth.text(1 << 20, seed=synth.SEED_BASE + 1)
    member = pygzip.compress(data.tobytes(), mtime=0)
    # the oracle (CPU restatement of gzip::Decoder) ...
    rc, out, used, _ = oracle.decode(oracle.GZIP, member)
    assert rc == 0 and out == data.tobytes() and used == len(member)
    # ... and the HIP path through the stream ABI, 8 KiB reads
    src = np.frombuffer(member, dtype=np.uint8)
    dec = np.zeros(data.size + 8192, dtype=np.uint8)
    rc, ol, _t = stream_copy.decode(ctx, ffi.GZIP, src.ctypes.data, src.size, 8192, dec.ctypes.data, dec.size)
    assert rc == 0 and ol == data.size and (dec[:ol] == data).all()
    # the same member through read sizes that do not divide anything
    for chunk in (1, 4093, 70001):
        if chunk == 1:
            small = pygzip.compress(data[:20000].tobytes(), mtime=0)
            s2 = np.frombuffer(small, dtype=np.uint8)
            rc, ol, _t = stream_copy.decode(ctx, ffi.GZIP, s2.ctypes.data, s2.size, 1, dec.ctypes.data, dec.size)
            assert rc == 0 and ol == 20000 and (dec[:ol] == data[:20000]).all()
        else:
            rc, ol, _t = stream_copy.decode(ctx, ffi.GZIP, src.ctypes.data, src.size, chunk, dec.ctypes.data, dec.size)
            assert rc == 0 and ol == data.size and (dec[:ol] == data).all()


@pytest.mark.parametrize("n", [0, 1, 70000, (2 << 20) - 1, 2 << 20, (9 << 20) + 3, 40 << 20])
def test_host_calls_pageable_and_pinned_vs_oracle(ctx, ffi, oracle, synth, n):
    """H2D / D2H by copy threads through page-locked slabs (pageable memory, from 2 MiB on) and by plain DMA (page-locked
    memory): same bytes either way, equal to the oracle's."""
    L = ffi.lib()
    data = synth.text(max(n, 1))[:n]
    opts, sched = ffi.make_opts(mtime=3), ffi.make_schedule(8192)
    want = oracle.encode(oracle.GZIP, data.tobytes(), write_size=8192, mtime=3)
    bound = L.lfx_encode_bound(n, C.byref(opts), C.byref(sched))
    out = np.zeros(bound, dtype=np.uint8)
    back = np.zeros(max(n, 1), dtype=np.uint8)
    m = ctx.encode_host_ptr(ffi.GZIP, data.ctypes.data if n else None, n, out.ctypes.data, bound, opts, sched)
    assert out[:m].tobytes() == want
    rc, ol, used, msg = ctx.decode_host_ptr(ffi.GZIP, out.ctypes.data, m, back.ctypes.data, n)
    assert (rc, ol, used) == (0, n, m), msg
    assert (back[:n] == data).all()
    pin_in, pin_out, pin_back = L.lfx_host_alloc(max(n, 1)), L.lfx_host_alloc(bound), L.lfx_host_alloc(max(n, 1))
    assert pin_in and pin_out and pin_back
    try:
        if n:
            C.memmove(pin_in, data.ctypes.data, n)
        m2 = ctx.encode_host_ptr(ffi.GZIP, pin_in, n, pin_out, bound, opts, sched)
        assert m2 == m and C.string_at(pin_out, m2) == want
        rc, ol, used, msg = ctx.decode_host_ptr(ffi.GZIP, pin_out, m2, pin_back, n)
        assert (rc, ol, used) == (0, n, m), msg
        assert C.string_at(pin_back, n) == data.tobytes()
        # a truncated member from host memory: the reference's error kind, output so far delivered
        if n >= 70000:
            rc, ol, used, msg = ctx.decode_host_ptr(ffi.GZIP, pin_out, m2 - 9, pin_back, n)
            assert rc == ffi.E_UNEXPECTED_EOF
    finally:
        for p in (pin_in, pin_out, pin_back):
            L.lfx_host_free(p)
    assert ctx.match_fallbacks() == 0


def test_io_copy_protocol_through_the_stream_abi_vs_oracle(ffi, lfx, oracle, synth, monkeypatch):
    """8192-byte write() calls into gzip::Encoder, 8192-byte read() calls out of gzip::Decoder (examples/flate.rs:52,96-97),
    from C; the stream encoder's batches (page-locked pending buffer, DMA out of it, output handed to the sink from
    page-locked staging) at three batch sizes — the write schedule, not the batching, decides the bytes."""
    import stream_copy
    n = 40 << 20
    data = synth.text(n, seed=synth.SEED_BASE + 9)
    want = oracle.encode(oracle.GZIP, data.tobytes(), write_size=8192, mtime=0)
    opts = ffi.make_opts(mtime=0)
    encb = np.zeros(n + n // 4 + 4096, dtype=np.uint8)
    dec = np.zeros(n, dtype=np.uint8)
    for batch_mb in (None, 1, 32):
        if batch_mb is None:
            monkeypatch.delenv("LFX_ENC_BATCH_MB", raising=False)
        else:
            monkeypatch.setenv("LFX_ENC_BATCH_MB", str(batch_mb))
        c2 = lfx.Context(0)                       # (diagnostic switches are read when a context is made)
        try:
            rc, m, _t = stream_copy.encode(c2, ffi.GZIP, opts, data.ctypes.data, n, 8192, encb.ctypes.data, encb.size)
            assert rc == 0 and encb[:m].tobytes() == want, (batch_mb, rc, m, len(want))
            dec[:] = 0
            rc, ol, _t = stream_copy.decode(c2, ffi.GZIP, encb.ctypes.data, m, 8192, dec.ctypes.data, n)
            assert rc == 0 and ol == n and (dec == data).all(), batch_mb
        finally:
            c2.close()
    # odd write sizes: another schedule, another stream — still the oracle's
    for ws in (1000, 65537):
        want2 = oracle.encode(oracle.ZLIB, data[:5 << 20].tobytes(), write_size=ws)
        c2 = lfx.Context(0)
        try:
            rc, m, _t = stream_copy.encode(c2, ffi.ZLIB, None, data.ctypes.data, 5 << 20, ws, encb.ctypes.data, encb.size)
            assert rc == 0 and encb[:m].tobytes() == want2, ws
        finally:
            c2.close()


def test_gzip_no_compression_then_header_keeps_the_headers_level(ctx, lfx):
    """EncodeOptions::no_compression() resets the header's level when it is called (gzip.rs:703), header(h)
    afterwards replaces the header — level included (gzip.rs:717-720): XFL 4 survives in that order, not in the other."""
    import io
    cloned = dict(modification_time=5, os=3, is_text=False, is_verified=False, extra_field=None, filename=None, comment=None, xfl=4)
    for opts, xfl in ((lfx.gzip.EncodeOptions().no_compression().header(cloned), 4),
                      (lfx.gzip.EncodeOptions().header(cloned).no_compression(), 0)):
        sink = io.BytesIO()
        e = lfx.gzip.Encoder.with_options(sink, opts, context=ctx)
        e.write(b"Hello World!")
        e.finish()
        b = sink.getvalue()
        assert b[:4] == b"\x1f\x8b\x08\x00" and b[4:8] == (5).to_bytes(4, "little") and b[8] == xfl
        assert pygzip.decompress(b) == b"Hello World!"


def test_single_pass_decode_of_large_blocks(ffi, lfx, oracle, synth, monkeypatch):
    """Round 6: for a stream's own large blocks the scan stores its code words per lane and blk_place_kernel moves them
    (one Huffman pass); blocks whose lanes overflow their regions — forced here by LFX_STORE_TIGHT — and everything under
    LFX_TWO_PASS take blk_emit_kernel as before.  All three give the input back, with libflate's error behaviour on damage
    (decode.rs:112-164: the bytes of the blocks in front of the damaged one, then InvalidData)."""
    n = 24 << 20
    data = synth.text(n, seed=synth.SEED_BASE + 11).tobytes()
    low = synth.lowent(8 << 20).tobytes()
    streams = {
        "text, 1 MiB blocks": oracle.encode(oracle.GZIP, data, write_size=8192, mtime=0),
        "text, 4 MiB blocks": oracle.encode(oracle.ZLIB, data, write_size=65536, block_size=4 << 20),
        "text, 300 KiB blocks": oracle.encode(oracle.DEFLATE, data[:8 << 20], write_size=1000, block_size=300 << 10),
        "low entropy (258-byte matches)": oracle.encode(oracle.ZLIB, low, write_size=8192),
        "fixed Huffman blocks": oracle.encode(oracle.GZIP, data[:6 << 20], write_size=8192, mtime=0, dynamic_huffman=0),
    }
    fmt = {"text, 1 MiB blocks": ffi.GZIP, "text, 4 MiB blocks": ffi.ZLIB, "text, 300 KiB blocks": ffi.DEFLATE,
           "low entropy (258-byte matches)": ffi.ZLIB, "fixed Huffman blocks": ffi.GZIP}
    want = {"text, 1 MiB blocks": data, "text, 4 MiB blocks": data, "text, 300 KiB blocks": data[:8 << 20],
            "low entropy (258-byte matches)": low, "fixed Huffman blocks": data[:6 << 20]}
    for env in ({}, {"LFX_STORE_TIGHT": "1"}, {"LFX_TWO_PASS": "1"}):
        for k in ("LFX_STORE_TIGHT", "LFX_TWO_PASS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c2 = lfx.Context(0)
        try:
            for name, z in streams.items():
                rc, out, used, msg = c2.decode_host(fmt[name], z)
                assert (rc, used) == (0, len(z)) and out == want[name], (env, name, rc, msg)
            # damage in the middle of the sixth block: same status, same bytes delivered as the oracle
            z = bytearray(streams["text, 1 MiB blocks"])
            z[len(z) * 6 // 24 + 1000] ^= 0x55
            rc, out, used, msg = c2.decode_host(ffi.GZIP, bytes(z))
            orc, oout, _oused, omsg = oracle.decode(oracle.GZIP, bytes(z))
            assert rc == orc and rc != 0 and out == oout, (env, rc, orc, len(out), len(oout), msg, omsg)
        finally:
            c2.close()


def test_stream_encoder_batch_in_flight_edge_cases(ctx, lfx, ffi, oracle, synth):
    """One batch in flight on the GPU (round 6): handles that share a context and take turns, a flush and a sync flush while a
    batch is in flight, a sink that fails when a batch is collected, an encoder dropped with a batch in flight."""
    import io
    data = synth.text(24 << 20, seed=synth.SEED_BASE + 12).tobytes()
    W = 1 << 20
    # (1) two encoders on ONE context, writes interleaved: each gets the oracle's bytes for its own schedule
    sa, sb = io.BytesIO(), io.BytesIO()
    ea = lfx.gzip.Encoder.new(sa, context=ctx)
    eb = lfx.zlib.Encoder.new(sb, context=ctx)
    for off in range(0, len(data), W):
        ea.write(data[off:off + W])
        eb.write(data[len(data) - off - W:len(data) - off])
    ea.finish()
    eb.finish()
    rev = b"".join(data[len(data) - off - W:len(data) - off] for off in range(0, len(data), W))
    assert sa.getvalue() == oracle.encode(oracle.GZIP, data, write_size=W, mtime=0)
    assert sb.getvalue() == oracle.encode(oracle.ZLIB, rev, write_size=W)
    # (2) flush() after 17 writes of 1 MiB (a batch is in flight, another block is open), then more writes — both zlib flush modes
    for sync in (0, 1):
        sink = io.BytesIO()
        opts = lfx.zlib.EncodeOptions()
        if sync:
            opts.flush_mode(lfx.zlib.FlushMode.SYNC)               # zlib.rs:504-507
        e = lfx.zlib.Encoder.with_options(sink, opts, context=ctx)
        o = oracle.Encoder(oracle.ZLIB, zlib_sync_flush=sync)
        for k, off in enumerate(range(0, len(data), W)):
            e.write(data[off:off + W]); o.write(data[off:off + W])
            if k == 16:
                e.flush(); o.flush()
                assert len(sink.getvalue()) > (6 << 20)            # everything written so far has reached the sink
        e.finish()
        assert sink.getvalue() == o.finish(), sync
    # (3) a sink that fails while a batch is being handed over: the collecting call reports it, later calls stay failed
    class Failing(io.BytesIO):
        def write(self, b):
            if self.tell() + len(b) > (3 << 20):
                raise OSError("disk full")
            return super().write(b)
    e = lfx.gzip.Encoder.new(Failing(), context=ctx)
    failed_at = None
    for k, off in enumerate(range(0, len(data), W)):
        try:
            e.write(data[off:off + W])
        except lfx.deflate.StreamError as err:
            failed_at = k
            assert err.status == ffi.E_IO
            break
    assert failed_at is not None and failed_at >= 15              # (the first batch's bytes arrive when the second is started)
    with pytest.raises(lfx.deflate.StreamError):
        e.write(b"more")
    e.into_inner()
    # (4) an encoder dropped with a batch in flight; the context encodes on
    e = lfx.gzip.Encoder.new(io.BytesIO(), context=ctx)
    for off in range(0, 9 << 20, W):
        e.write(data[off:off + W])
    e.into_inner()
    assert enc(ctx, ffi, ffi.GZIP, data[:3 << 20], 8192, mtime=0) == oracle.encode(oracle.GZIP, data[:3 << 20], write_size=8192, mtime=0)


def test_stream_decoder_decoding_ahead_edge_cases(ctx, lfx, ffi, oracle, synth):
    """The next window decoded by a worker thread while the caller drains the current one (round 6): members that span several
    windows in a MultiDecoder, damage and truncation inside a later window (libflate's error kind; the bytes of the blocks in
    front of it delivered first, decode.rs:136-164), a decoder freed while a window is in flight."""
    import io
    a = synth.text(40 << 20, seed=synth.SEED_BASE + 13).tobytes()
    b = synth.lowent(24 << 20).tobytes()
    za = oracle.encode(oracle.GZIP, a, write_size=8192, mtime=0)
    zb = oracle.encode(oracle.GZIP, b, write_size=8192, mtime=1)
    # (1) two members, read in 1 MiB pieces; trailing bytes behind the second are not a member
    d = lfx.gzip.MultiDecoder.new(io.BytesIO(za + zb), context=ctx)
    got = bytearray()
    while True:
        piece = d.read(1 << 20)
        if not piece:
            break
        got += piece
    assert bytes(got) == a + b
    # (2) a flipped byte 70 % into the member (a later window) / the member cut there: same status and bytes as the oracle
    for kind in ("flip", "cut"):
        z = bytearray(za)
        at = len(z) * 7 // 10
        if kind == "flip":
            z[at] ^= 0x10
        else:
            del z[at:]
        orc, oout, _used, _msg = oracle.decode(oracle.GZIP, bytes(z))
        assert orc != 0
        dd = lfx.gzip.Decoder.new(io.BytesIO(bytes(z)), context=ctx)
        out = bytearray()
        status = 0
        try:
            while True:
                piece = dd.read(4 << 20)
                if not piece:
                    break
                out += piece
        except lfx.deflate.StreamError as err:
            status = err.status
            out += getattr(err, "partial", b"")
        assert status == orc, (kind, status, orc)
        assert bytes(out) == oout[:len(out)] and len(out) <= len(oout), kind
        assert len(oout) - len(out) <= (2 << 20), (kind, len(out), len(oout))      # at most the damaged block's own bytes are withheld
    # (3) freed in the middle of a window, with the next one in flight; the context decodes on
    dd = lfx.gzip.Decoder.new(io.BytesIO(za), context=ctx)
    first = dd.read(100)
    assert first == a[:100]
    more = dd.read(5 << 20)            # (into the 16 MiB window: its successor is being decoded)
    assert more == a[100:100 + len(more)]
    del dd
    rc, out, used, _ = ctx.decode_host(ffi.GZIP, zb)
    assert rc == 0 and out == b and used == len(zb)


def test_small_blocks_on_256_lanes(ffi, lfx, oracle, synth, monkeypatch):
    """Round 6: blocks of a few tens of KB (another encoder's; the batch path's 64 KiB streams) are scanned and emitted by the
    256-lane instances of the kernels; LFX_NO_SMALL_SCAN keeps 1024 lanes.  Same bytes, same verdicts on damage, single
    stream and batch."""
    import zlib
    text = synth.text(12 << 20, seed=synth.SEED_BASE + 14).tobytes()
    low = synth.lowent(4 << 20).tobytes()
    rnd = np.random.default_rng(3).integers(0, 256, 2 << 20, dtype=np.uint8).tobytes()
    singles = {
        "zlib level 6": (ffi.ZLIB, zlib.compress(text, 6), text),
        "zlib level 1": (ffi.ZLIB, zlib.compress(text[:5 << 20] + low, 1), text[:5 << 20] + low),
        "zlib level 9 + random": (ffi.ZLIB, zlib.compress(text[:3 << 20] + rnd, 9), text[:3 << 20] + rnd),
        "reference, 64 KiB blocks": (ffi.GZIP, oracle.encode(oracle.GZIP, text[:6 << 20], write_size=4096, block_size=65536, mtime=0), text[:6 << 20]),
        "reference, 20 KB blocks, fixed codes": (ffi.DEFLATE, oracle.encode(oracle.DEFLATE, text[:2 << 20], write_size=1000, block_size=20000, dynamic_huffman=0), text[:2 << 20]),
    }
    count, size = 180, 65536
    streams = [zlib.compress(text[i * size:(i + 1) * size], 6) if i & 1 else oracle.encode(oracle.ZLIB, text[i * size:(i + 1) * size], write_size=0)
               for i in range(count)]
    streams[7] = streams[7][:len(streams[7]) // 2]                       # truncated
    bad = bytearray(streams[8]); bad[len(bad) // 2] ^= 0x20; streams[8] = bytes(bad)      # damaged
    L = ffi.lib()
    results = {}
    for env in ({}, {"LFX_NO_SMALL_SCAN": "1"}):
        monkeypatch.delenv("LFX_NO_SMALL_SCAN", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c2 = lfx.Context(0)
        try:
            for name, (fmt, z, want) in singles.items():
                rc, out, used, msg = c2.decode_host(fmt, z)
                assert (rc, used) == (0, len(z)) and out == want, (env, name, rc, msg)
                zz = bytearray(z); zz[len(zz) // 3] ^= 0x44
                rc, out, used, msg = c2.decode_host(fmt, bytes(zz))
                ofmt = {ffi.ZLIB: oracle.ZLIB, ffi.GZIP: oracle.GZIP, ffi.DEFLATE: oracle.DEFLATE}[fmt]
                orc, oout, _u, _m = oracle.decode(ofmt, bytes(zz))
                assert rc == orc and out == oout, (env, name, "damaged", rc, orc, len(out), len(oout))
            # the batch call
            import torch
            blob = b"".join(streams)
            offs = np.cumsum([0] + [len(x) for x in streams[:-1]]).astype(np.uint64)
            lens = np.array([len(x) for x in streams], dtype=np.uint64)
            d_in = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).cuda()
            d_out = torch.zeros(count * size, dtype=torch.uint8, device="cuda")
            out_off = (np.arange(count, dtype=np.uint64) * np.uint64(size))
            out_cap = np.full(count, size, dtype=np.uint64)
            out_len = np.zeros(count, dtype=np.uint64)
            status = np.zeros(count, dtype=np.int32)
            rc = L.lfx_decode_batch_device(c2.handle, ffi.ZLIB, count, d_in.data_ptr(), offs.ctypes.data, lens.ctypes.data, d_out.data_ptr(),
                                           out_off.ctypes.data, out_cap.ctypes.data, out_len.ctypes.data, status.ctypes.data)
            host = d_out.cpu().numpy()
            for i in range(count):
                if i in (7, 8):
                    orc, oout, _u, _m = oracle.decode(oracle.ZLIB, streams[i])
                    assert status[i] == orc and orc != 0, (env, i, status[i], orc)
                else:
                    assert status[i] == 0 and out_len[i] == size and host[i * size:(i + 1) * size].tobytes() == text[i * size:(i + 1) * size], (env, i)
            results[bool(env)] = (status.copy(), out_len.copy())
        finally:
            c2.close()
    assert (results[False][0] == results[True][0]).all() and (results[False][1] == results[True][1]).all()


def test_histogram_inside_the_parse_and_by_its_own_kernel(ffi, lfx, oracle, synth, monkeypatch):
    """Round 6: the blocks' symbol counts (DynamicHuffmanCodec::build, symbol.rs:320-341) are taken by the kernel that writes
    the code words (parse_emit_hist_kernel + the chunk tails and EndOfBlock in parse_fix_kernel); LFX_HIST_SEPARATE=1 keeps
    histogram_kernel.  Both give the oracle's bytes on every chunk shape: 256 KiB chunks, one huge chunk (S1), ragged write
    lists with flushes, literal-only chunks, a last block of three bytes, many small chunks."""
    text = synth.text(9 << 20, seed=synth.SEED_BASE + 21).tobytes()
    low = synth.lowent(3 << 20).tobytes()
    cases = [
        (ffi.GZIP, oracle.GZIP, text, dict(write_size=8192), dict(mtime=0)),
        (ffi.ZLIB, oracle.ZLIB, text[: (3 << 20) + 3], dict(write_size=0), {}),                                   # one chunk
        (ffi.DEFLATE, oracle.DEFLATE, text[: (1 << 20) + 3], dict(write_size=8192), {}),                            # a 3-byte tail block
        (ffi.ZLIB, oracle.ZLIB, low, dict(write_size=8192), {}),
        (ffi.GZIP, oracle.GZIP, text[: 2 << 20], dict(write_size=8192), dict(mtime=0, no_compression=1)),
        (ffi.DEFLATE, oracle.DEFLATE, text[: 3 << 20], dict(write_size=1000), dict(block_size=300 << 10)),
        (ffi.ZLIB, oracle.ZLIB, text[: 2 << 20], dict(write_size=8192), dict(lz77_kind=1)),                         # NoCompressionLz77Encoder: literals
        (ffi.DEFLATE, oracle.DEFLATE, b"", dict(write_size=8192), {}),
        (ffi.DEFLATE, oracle.DEFLATE, b"abc", dict(write_size=8192), {}),
    ]
    want = []
    for _f, of, d, sk, ok in cases:
        try:
            want.append(oracle.encode(of, d, **sk, **ok))
        except TypeError:
            want.append(None)
    for env in ({}, {"LFX_HIST_SEPARATE": "1"}):
        monkeypatch.delenv("LFX_HIST_SEPARATE", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c2 = lfx.Context(0)
        try:
            for (f, _of, d, sk, ok), w in zip(cases, want):
                if w is None:
                    continue
                got = c2.encode_host(f, d, ffi.make_opts(**ok), ffi.make_schedule(sk["write_size"]))
                assert got == w, (env, f, len(d), sk, ok)
            # a ragged list of writes with flushes: chunks of every size, blocks that end inside what a segment group covers
            rng = np.random.default_rng(5)
            writes, left = [], len(text)
            while left:
                w = int(min(left, rng.choice([1, 100, 8192, 70000, 300000, 1 << 20])))
                writes.append(w)
                left -= w
                if rng.random() < 0.2:
                    writes.append(None)
            got = c2.encode_host(ffi.ZLIB, text, ffi.make_opts(), ffi.make_schedule(0, writes))
            rc, out, used, msg = c2.decode_host(ffi.ZLIB, got)
            assert (rc, used) == (0, len(got)) and out == text, (env, rc, msg)
        finally:
            c2.close()
    first = {}
    for env in ("0", "1"):      # ... and the two agree with each other on the ragged list
        monkeypatch.setenv("LFX_HIST_SEPARATE", env)
        c2 = lfx.Context(0)
        try:
            first[env] = c2.encode_host(ffi.ZLIB, text, ffi.make_opts(), ffi.make_schedule(0, writes))
        finally:
            c2.close()
    assert first["0"] == first["1"]


def test_finder_reports_every_dynamic_header(ffi, lfx, oracle, synth, monkeypatch, capfd):
    """The finder's stage 2 walks a candidate's literal / length widths in a lean loop and the rest in the general one
    (round 6).  A header it failed to report would not fail a decode — the chain walk scans such a block on demand, a launch
    later — so the count is checked: LFX_DEBUG's `candidates` = the stream's blocks, for 1 MiB blocks (HLIT / HDIST of text)
    and for small blocks of low-entropy data (short code-length sequences)."""
    import re
    text = synth.text(20 << 20, seed=synth.SEED_BASE + 22).tobytes()
    low = synth.lowent(6 << 20).tobytes()
    streams = [
        (ffi.GZIP, oracle.encode(oracle.GZIP, text, write_size=8192, mtime=0), text, 20),
        (ffi.DEFLATE, oracle.encode(oracle.DEFLATE, text[: 12 << 20], write_size=1000, block_size=300 << 10), text[: 12 << 20], 40),
        (ffi.ZLIB, oracle.encode(oracle.ZLIB, low, write_size=8192, block_size=256 << 10), low, 24),
    ]
    monkeypatch.setenv("LFX_DEBUG", "1")
    c2 = lfx.Context(0)
    try:
        for f, z, want, nblocks in streams:
            capfd.readouterr()
            rc, out, used, msg = c2.decode_host(f, z)
            err = capfd.readouterr().err
            assert (rc, used) == (0, len(z)) and out == want, (rc, msg)
            # (a stream of few blocks is scanned in pieces between the finder's candidates: that line carries the count then)
            m = re.search(r"finder: stage1=\d+ candidates=(\d+) scan jobs=\d+", err) or re.search(r"pieces over (\d+) candidate ranges", err)
            assert m, err[-2000:]
            assert nblocks <= int(m.group(1)) <= nblocks + 2, (nblocks, m.group(0))
            m2 = re.search(r"chain ok=1 blocks=(\d+) .* on_demand=(\d+)", err)
            assert m2 and int(m2.group(1)) >= nblocks and int(m2.group(2)) <= 1, err[-2000:]     # (at most the final block)
    finally:
        c2.close()


def test_decode_paths_of_the_second_half_of_round_6(ffi, lfx, oracle, synth, monkeypatch):
    """The decode's small transfers through page-locked slots (Ctx::small_up / small_down) and as the plain pageable copies a full
    arena falls back to (LFX_NO_PIN_SLOTS=1); a cut-down finder stage 2 in front of the real one (LFX_FIND2_EXP, timing only) must
    not change an answer; candidates at the very end of a stream (the empty final block of the reference; a stream cut inside its
    last header) take the staged walk with clamped loads: same bytes, same verdicts as the oracle."""
    text = synth.text(12 << 20, seed=synth.SEED_BASE + 31).tobytes()
    z_ref = oracle.encode(oracle.GZIP, text, write_size=8192, mtime=0)
    z_small_blocks = oracle.encode(oracle.DEFLATE, text[: 6 << 20], write_size=1000, block_size=100 << 10)
    import zlib as pyzlib
    z_py = pyzlib.compress(text, 6)
    cases = [(ffi.GZIP, oracle.GZIP, z_ref, text), (ffi.DEFLATE, oracle.DEFLATE, z_small_blocks, text[: 6 << 20]), (ffi.ZLIB, oracle.ZLIB, z_py, text)]
    for env in ({}, {"LFX_NO_PIN_SLOTS": "1"}, {"LFX_FIND2_EXP": "1"}, {"LFX_FIND2_EXP": "4"}):
        for k in ("LFX_NO_PIN_SLOTS", "LFX_FIND2_EXP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c2 = lfx.Context(0)
        try:
            for f, of, z, want in cases:
                rc, out, used, msg = c2.decode_host(f, z)
                assert (rc, used) == (0, len(z)) and out == want, (env, f, rc, msg)
            # cut inside the last blocks' headers and right behind them: the oracle's verdict and bytes
            for cut in (len(z_small_blocks) - 1, len(z_small_blocks) - 3, len(z_small_blocks) - 40, len(z_small_blocks) - 700):
                zc = z_small_blocks[:cut]
                rc, out, used, msg = c2.decode_host(ffi.DEFLATE, zc)
                orc, oout, _ou, omsg = oracle.decode(oracle.DEFLATE, zc)
                assert rc == orc and out == oout, (env, cut, rc, orc, len(out), len(oout), msg, omsg)
        finally:
            c2.close()


def test_context_outlives_its_handles_whatever_the_finalizer_order(lfx, oracle, synth):
    """A native handle uses its context until it is freed (lfx_*_free takes the context's mutex).  The garbage collector runs the
    finalizers of an unreachable group in an undefined order — at interpreter exit Context.__del__ ran before a decoder's, and
    the decoder's free then worked on freed memory (an abort under MALLOC_PERTURB_, a corrupted heap without).  The wrappers
    count themselves in and out of their Context: closing a context that still has handles frees nothing until the last one
    is gone."""
    import gc
    import io
    data = synth.text(300000).tobytes()
    stream = oracle.encode(oracle.GZIP, data, 8192)
    own = lfx.Context(0)
    d = lfx.gzip.Decoder.new(stream, context=own)
    sink = io.BytesIO()
    e = lfx.gzip.Encoder.new(sink, context=own)
    z = lfx.lz77.DefaultLz77Encoder(context=own)
    assert own._users == 3
    own.close()                                  # (what Context.__del__ does)
    assert own.handle                            # still there: three handles use it
    assert d.read_to_end() == data               # ... and it works
    e.write(data[:100000])
    e.finish()                                   # frees the encoder's handle
    assert own._users == 2 and own.handle
    del d
    gc.collect()
    assert own._users == 1 and own.handle
    del z
    gc.collect()
    assert own._users == 0 and not own.handle    # the last handle took the context with it
    assert sink.getvalue() == oracle.encode(oracle.GZIP, data[:100000], 0)
