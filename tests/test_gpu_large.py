"""GPU: BASELINE.json-size checks through size-independent properties (round trip, checksum of the
round trip, python-zlib as an independent inflater) plus a 64 MiB byte-for-byte oracle comparison."""
import os
import zlib as pyzlib

import numpy as np
import pytest

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
    return libflate_amd.Context(0), _ffi, synth


def test_cfg2_64mib_bit_exact_vs_oracle(env, oracle):
    import torch
    ctx, ffi, synth = env
    n = 64 << 20
    data = synth.text(n)
    d_in = torch.from_numpy(data).cuda()
    sched, opts = ffi.make_schedule(8192), ffi.make_opts()
    bound = ffi.lib().lfx_encode_bound(n, None, None)
    d_out = torch.empty(bound, dtype=torch.uint8, device="cuda")
    m = ctx.encode_device(ffi.GZIP, d_in.data_ptr(), n, d_out.data_ptr(), bound, opts, sched)
    got = d_out[:m].cpu().numpy().tobytes()
    want = oracle.encode(oracle.GZIP, data.tobytes(), write_size=8192)
    assert got == want
    assert len(oracle.scan_blocks(got[10:-8])) == 65   # 64 one-MiB blocks + the empty final block


def test_cfg2_256mib_roundtrip(env):
    import torch
    ctx, ffi, synth = env
    n = 256 << 20
    data = synth.text(n)
    d_in = torch.from_numpy(data).cuda()
    sched, opts = ffi.make_schedule(8192), ffi.make_opts()
    bound = ffi.lib().lfx_encode_bound(n, None, None)
    d_out = torch.empty(bound, dtype=torch.uint8, device="cuda")
    m = ctx.encode_device(ffi.GZIP, d_in.data_ptr(), n, d_out.data_ptr(), bound, opts, sched)
    comp = d_out[:m].cpu().numpy().tobytes()
    # independent inflater + independent CRC
    plain = pyzlib.decompress(comp, 31)
    assert len(plain) == n and pyzlib.crc32(plain) == pyzlib.crc32(data.tobytes())
    assert int.from_bytes(comp[-8:-4], "little") == pyzlib.crc32(plain) and int.from_bytes(comp[-4:], "little") == n
    d_dec = torch.zeros(n, dtype=torch.uint8, device="cuda")
    rc, ol, used, msg = ctx.decode_device(ffi.GZIP, d_out.data_ptr(), m, d_dec.data_ptr(), n)
    assert (rc, ol, used) == (0, n, m), msg
    assert torch.equal(d_dec, d_in)


def test_cfg5_lowent_zlib(env, oracle):
    import torch
    ctx, ffi, synth = env
    n = 32 << 20                               # the oracle finishes this in seconds; cfg5 proper is 1 GiB
    data = synth.lowent(n)
    d_in = torch.from_numpy(data).cuda()
    sched, opts = ffi.make_schedule(8192), ffi.make_opts()
    bound = ffi.lib().lfx_encode_bound(n, None, None)
    d_out = torch.empty(bound, dtype=torch.uint8, device="cuda")
    m = ctx.encode_device(ffi.ZLIB, d_in.data_ptr(), n, d_out.data_ptr(), bound, opts, sched)
    got = d_out[:m].cpu().numpy().tobytes()
    assert got == oracle.encode(oracle.ZLIB, data.tobytes(), write_size=8192)
    d_dec = torch.zeros(n, dtype=torch.uint8, device="cuda")
    rc, ol, used, msg = ctx.decode_device(ffi.ZLIB, d_out.data_ptr(), m, d_dec.data_ptr(), n)
    assert (rc, ol, used) == (0, n, m), msg
    assert torch.equal(d_dec, d_in)


def test_s1_single_write_multi_segment(env, oracle):
    # schedule S1 on 3 MiB: ONE LZ77 chunk split into 256 Ki-position segments with 32 KiB warm-up
    ctx, ffi, synth = env
    data = synth.text(3 << 20).tobytes()
    got = ctx.encode_host(ffi.GZIP, data, ffi.make_opts(), ffi.make_schedule(0))
    assert got == oracle.encode(oracle.GZIP, data, write_size=0)
    assert ctx.decode_host(ffi.GZIP, got)[:2] == (0, data)


def test_cfg3_batch_4096_streams(env, oracle):
    """BASELINE cfg3 at full size: 4096 independent 64 KiB zlib streams, half made by the restated reference
    encoder (one dynamic block + the empty final block each), half by python zlib level 6 (foreign encoder);
    bytes and Adler-32 verified (a checksum mismatch would surface as a status)."""
    import time
    import torch
    ctx, ffi, synth = env
    count, size = 4096, 65536
    big = synth.text(count * size, seed=0x5EED0003)
    plains = [big[i * size:(i + 1) * size].tobytes() for i in range(count)]
    streams = [oracle.encode(oracle.ZLIB, p) if i % 2 else pyzlib.compress(p, 6) for i, p in enumerate(plains)]
    blob = b"".join(streams)
    in_len = np.array([len(s) for s in streams], dtype=np.uint64)
    in_off = (np.cumsum(in_len) - in_len).astype(np.uint64)
    out_off = (np.arange(count, dtype=np.uint64) * size)
    out_cap = np.full(count, size, dtype=np.uint64)
    d_in = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
    d_out = torch.zeros(count * size, dtype=torch.uint8, device="cuda")
    out_len = np.zeros(count, dtype=np.uint64)
    status = np.zeros(count, dtype=np.int32)
    ctx.enable_timing(True)
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = ffi.lib().lfx_decode_batch_device(ctx.handle, ffi.ZLIB, count, d_in.data_ptr(), in_off.ctypes.data,
                                               in_len.ctypes.data, d_out.data_ptr(), out_off.ctypes.data,
                                               out_cap.ctypes.data, out_len.ctypes.data, status.ctypes.data)
        dt = time.perf_counter() - t0
    assert rc == 0
    assert not status.any() and (out_len == size).all()
    assert torch.equal(d_out, torch.from_numpy(big).cuda())
    print("cfg3: %d streams, %.1f ms, %.2f GB/s of output" % (count, dt * 1e3, count * size / dt / 1e9), ctx.last_timing())


def test_foreign_streams_cross_block(env):
    """Streams of another encoder (python zlib, levels 1 / 6 / 9, and python gzip): their blocks read the output
    of earlier blocks, which the reference's own blocks never do.  They must stay on the GPU's block-parallel
    scan/emit path (ordered materialisation), not fall back to the serial kernel."""
    import gzip as pygzip
    import os
    import torch
    ctx, ffi, synth = env
    n = 8 << 20
    data = synth.text(n, seed=0x5EED0007)
    raw = data.tobytes()
    d_want = torch.from_numpy(data).cuda()
    os.environ["LFX_NO_SERIAL"] = "1"
    try:
        for fmt, comp in ((ffi.ZLIB, pyzlib.compress(raw, 1)), (ffi.ZLIB, pyzlib.compress(raw, 6)),
                          (ffi.ZLIB, pyzlib.compress(raw, 9)), (ffi.GZIP, pygzip.compress(raw, 6, mtime=0))):
            d_in = torch.frombuffer(bytearray(comp), dtype=torch.uint8).cuda()
            d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
            rc, ol, used, msg = ctx.decode_device(fmt, d_in.data_ptr(), len(comp), d_out.data_ptr(), n)
            assert (rc, ol, used) == (0, n, len(comp)), msg
            assert torch.equal(d_out, d_want)
    finally:
        os.environ.pop("LFX_NO_SERIAL", None)


def test_foreign_streams_stored_and_fixed_blocks(env):
    """Blocks the finder cannot see (it looks for dynamic-block headers): sync-flush markers (empty stored
    blocks) inside a stream, an all-stored stream and a short fixed-Huffman stream are scanned on demand while
    the chain is walked — still no serial kernel."""
    import os
    import torch
    ctx, ffi, synth = env
    raw = synth.text(6 << 20, seed=0x5EED0008).tobytes()
    streams = []
    co = pyzlib.compressobj(6)
    parts = []
    for i in range(0, len(raw), 1 << 20):
        parts.append(co.compress(raw[i:i + (1 << 20)]))
        parts.append(co.flush(pyzlib.Z_SYNC_FLUSH))
    parts.append(co.flush())
    streams.append((raw, b"".join(parts)))
    streams.append((raw[:2 << 20], pyzlib.compress(raw[:2 << 20], 0)))                       # stored blocks only
    cf = pyzlib.compressobj(6, pyzlib.DEFLATED, 15, 8, pyzlib.Z_FIXED)
    streams.append((raw[:300000], cf.compress(raw[:300000]) + cf.flush()))                  # fixed-Huffman blocks
    os.environ["LFX_NO_SERIAL"] = "1"
    try:
        for plain, comp in streams:
            n = len(plain)
            d_in = torch.frombuffer(bytearray(comp), dtype=torch.uint8).cuda()
            d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
            rc, ol, used, msg = ctx.decode_device(ffi.ZLIB, d_in.data_ptr(), len(comp), d_out.data_ptr(), n)
            assert (rc, ol, used) == (0, n, len(comp)), (msg, len(comp))
            assert d_out.cpu().numpy().tobytes() == plain
    finally:
        os.environ.pop("LFX_NO_SERIAL", None)


def test_s1_huge_block_pieces(env, oracle):
    """Schedule S1 on 24 MiB: ONE LZ77 chunk and ONE block.  Encode is compared with the oracle byte for byte;
    decode must scan the block in pieces (a proven chain of workgroups) and materialise it through markers —
    no serial kernel."""
    import os
    import torch
    ctx, ffi, synth = env
    n = 24 << 20
    data = synth.text(n, seed=0x5EED0009)
    d_in = torch.from_numpy(data).cuda()
    sched, opts = ffi.make_schedule(0), ffi.make_opts()
    bound = ffi.lib().lfx_encode_bound(n, None, None)
    d_out = torch.empty(bound, dtype=torch.uint8, device="cuda")
    m = ctx.encode_device(ffi.GZIP, d_in.data_ptr(), n, d_out.data_ptr(), bound, opts, sched)
    got = d_out[:m].cpu().numpy().tobytes()
    assert got == oracle.encode(oracle.GZIP, data.tobytes(), write_size=0)
    assert len(oracle.scan_blocks(got[10:-8])) == 2          # the block + the empty final block
    os.environ["LFX_NO_SERIAL"] = "1"
    try:
        d_dec = torch.zeros(n, dtype=torch.uint8, device="cuda")
        rc, ol, used, msg = ctx.decode_device(ffi.GZIP, d_out.data_ptr(), m, d_dec.data_ptr(), n)
        assert (rc, ol, used) == (0, n, m), msg
        assert torch.equal(d_dec, d_in)
    finally:
        os.environ.pop("LFX_NO_SERIAL", None)


def test_s1_huge_block_corrupted(env, oracle):
    """A damaged huge-block stream: the piece chain (or the scan behind it) refuses it and the exact serial kernel
    must report what the reference reports — same status, same bytes produced so far."""
    ctx, ffi, synth = env
    n = 12 << 20
    data = synth.text(n, seed=0x5EED000A).tobytes()
    good = oracle.encode(oracle.ZLIB, data, write_size=0)
    assert len(good) > (4 << 20)
    bad = bytearray(good)
    bad[len(bad) * 3 // 5] ^= 0x10
    want = oracle.decode(oracle.ZLIB, bytes(bad))
    rc, out, used, msg = ctx.decode_host(ffi.ZLIB, bytes(bad))
    assert rc == want[0] and rc != 0
    assert out == want[1]
    # truncation
    cut = bytes(good[:len(good) * 2 // 3])
    want = oracle.decode(oracle.ZLIB, cut)
    rc, out, used, msg = ctx.decode_host(ffi.ZLIB, cut)
    assert rc == want[0] == ffi.E_UNEXPECTED_EOF
    assert out == want[1]


def test_foreign_streams_fuzz(env):
    """python zlib as a stand-in for "any other encoder": levels, strategies, window sizes and flush points chosen
    at random; every stream must decode to its input (whichever internal path it takes)."""
    import torch
    ctx, ffi, synth = env
    rng = np.random.default_rng(77)
    text = synth.text(6 << 20, seed=0x5EED000B).tobytes()
    low = synth.lowent(3 << 20, seed=0x5EED000C).tobytes()
    strategies = [pyzlib.Z_DEFAULT_STRATEGY, pyzlib.Z_FILTERED, pyzlib.Z_HUFFMAN_ONLY, pyzlib.Z_RLE, pyzlib.Z_FIXED]
    for trial in range(int(os.environ.get("LFX_FOREIGN_TRIALS", "24"))):
        src = text if trial % 3 else low
        n = int(rng.integers(300000, min(len(src), 4 << 20)))
        o = int(rng.integers(0, len(src) - n + 1))
        plain = src[o:o + n]
        level = int(rng.integers(0, 10))
        wbits = int(rng.integers(9, 16))
        strat = strategies[int(rng.integers(0, len(strategies)))]
        co = pyzlib.compressobj(level, pyzlib.DEFLATED, wbits, 8, strat)
        parts, at = [], 0
        while at < n:
            step = int(rng.integers(20000, 900000))
            parts.append(co.compress(plain[at:at + step]))
            at += step
            r = int(rng.integers(0, 6))
            if r == 0: parts.append(co.flush(pyzlib.Z_SYNC_FLUSH))
            if r == 1: parts.append(co.flush(pyzlib.Z_FULL_FLUSH))
        parts.append(co.flush())
        comp = b"".join(parts)
        d_in = torch.frombuffer(bytearray(comp), dtype=torch.uint8).cuda()
        d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
        rc, ol, used, msg = ctx.decode_device(ffi.ZLIB, d_in.data_ptr(), len(comp), d_out.data_ptr(), n)
        assert (rc, ol, used) == (0, n, len(comp)), (trial, level, wbits, strat, msg)
        assert d_out.cpu().numpy().tobytes() == plain, (trial, level, wbits, strat)
