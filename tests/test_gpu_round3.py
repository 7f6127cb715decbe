"""GPU: round-3 parity cases — the lazy-length parse (lfx_parse2.hip) at its group / segment / workgroup boundaries and
on data that drives its repair paths, the first-generation match kernel behind LFX_MATCH_V1 (the fallback the host takes
on a lane-order violation), and the hardware property the head pass of lfx_match3.hip rests on."""
import os
import subprocess
import sys

import numpy as np
import pytest

from golden import kat

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


def corpus(synth):
    rng = np.random.default_rng(11)
    text = synth.text(1 << 20).tobytes()
    return {
        "text1m": text,
        "zeros": bytes(300000),
        "lowent": synth.lowent(700000).tobytes(),
        "abc": b"abc" * 100000,
        "period7": b"abcdefg" * 40000,
        "rand4": rng.integers(0, 4, 200000, dtype=np.uint8).tobytes(),
        "rand": rng.integers(0, 256, 100000, dtype=np.uint8).tobytes(),
    }


# ------------------------------------------------------------------ a-2 / a-3: the walk with lazy lengths
def test_lazy_parse_boundaries_vs_oracle(env, oracle):
    """PARSE_GROUP = 52 positions per lane, 64 lanes per segment (3328), four segments per workgroup (13312): every
    size around those boundaries, through both schedules (S1: one chunk; 8192-byte writes: 256 KiB chunks)."""
    lfx, ctx, ffi, synth = env
    text = synth.text(70000).tobytes()
    low = synth.lowent(70000).tobytes()
    sizes = [0, 1, 2, 3, 4, 5, 6, 51, 52, 53, 54, 55, 56, 103, 104, 105, 3326, 3327, 3328, 3329, 3330, 3331, 3332,
             6655, 6656, 6657, 6659, 13311, 13312, 13313, 13315, 13316, 26624, 26627, 39936, 46080, 66560, 66563]
    for n in sizes:
        for src in (text, low):
            data = src[:n]
            for ws in (0, 8192):
                got = ctx.encode_host(ffi.DEFLATE, data, ffi.make_opts(), ffi.make_schedule(ws))
                assert got == oracle.encode(oracle.DEFLATE, data, write_size=ws), (n, ws, src is low)


def test_lazy_parse_repair_paths_vs_oracle(env, oracle):
    """Runs and periodic data: matches of 258 jump over whole groups, speculative group walks never merge with the true
    walk, segments are entered far from where they were assumed — the serial repair inside parse_walk_kernel and the
    re-walks of parse_fixseg / parse_fix through global memory.  Window / length limits narrower than the defaults
    (default.rs:226-243) go through the same code."""
    lfx, ctx, ffi, synth = env
    for name, data in corpus(synth).items():
        for ws in (0, 8192):
            got = ctx.encode_host(ffi.ZLIB, data, ffi.make_opts(), ffi.make_schedule(ws))
            assert got == oracle.encode(oracle.ZLIB, data, write_size=ws), (name, ws)
    data = synth.text(300000).tobytes()
    for window, max_len in ((1024, 20), (32768, 3), (100, 258), (32768, 4)):
        for src in (data, synth.lowent(200000).tobytes()):
            got = ctx.encode_host(ffi.DEFLATE, src, ffi.make_opts(window_size=window, max_length=max_len), ffi.make_schedule(8192))
            assert got == oracle.encode(oracle.DEFLATE, src, write_size=8192, window_size=window, max_length=max_len), (window, max_len)


def test_lz77_plugin_codes_vs_oracle(env, oracle):
    """The Lz77Encode plug-in (libflate_lz77/src/lib.rs:83-107) hands out the walk's code words themselves."""
    lfx, ctx, ffi, synth = env
    from libflate_amd import lz77
    for name, data in corpus(synth).items():
        data = data[:200000]                              # (below window * 8: one flush unit, default.rs:65)
        enc = lz77.DefaultLz77Encoder(context=ctx)
        codes = []
        enc.encode(data, codes)
        enc.flush(codes)
        want = [lz77.Code.from_word(w) for w in oracle.lz77_chunk(data)]
        assert codes == want, name


# ------------------------------------------------------------------ the fallback behind a lane-order violation
def test_match_v1_fallback_vs_oracle(env, oracle, monkeypatch):
    """lfx_api.cpp re-runs the first-generation match kernel (+ md → cd) when lfx_match3.hip reports a lane-order
    violation; LFX_MATCH_V1 selects it from the start (read once, when the context is created).  Same bytes as the
    oracle, through the same parse."""
    lfx, ctx, ffi, synth = env
    monkeypatch.setenv("LFX_MATCH_V1", "1")
    c1 = lfx.Context(0)
    monkeypatch.delenv("LFX_MATCH_V1")
    cases = dict(corpus(synth))
    cases["text3m_S1"] = synth.text(3 << 20).tobytes()
    for name, data in cases.items():
        ws = 0 if name.endswith("_S1") else 8192
        got = c1.encode_host(ffi.GZIP, data, ffi.make_opts(mtime=0), ffi.make_schedule(ws))
        assert got == oracle.encode(oracle.GZIP, data, write_size=ws, mtime=0), name
        assert got == ctx.encode_host(ffi.GZIP, data, ffi.make_opts(mtime=0), ffi.make_schedule(ws)), name
    c1.enable_timing(True)
    c1.encode_host(ffi.GZIP, cases["text1m"], ffi.make_opts(mtime=0), ffi.make_schedule(8192))
    c1.close()


def test_mskor_lane_order_property():
    """ds_mskor_rtn_b32 serves the lanes of one instruction that hit the same 16-bit field in ascending lane order, and a
    wavefront's instructions in issue order — the one hardware assumption of the head pass (lfx_match3.hip).  The
    microbenchmark is built and run here; its output is kept as profiles/r03_mskor_order.txt."""
    src = os.path.join(ROOT, "tools", "exp", "mskor_test.hip")
    exe = os.path.join(ROOT, "tools", "exp", "mskor_test")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-o", exe, src])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("mskor nbuckets=")]
    assert len(lines) == 7, out.stdout
    for l in lines:
        assert "violations=0 of 192000" in l, l
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r3_mskor_order.txt"), "w") as f:
        f.write("# tools/exp/mskor_test (built and run by tests/test_gpu_round3.py::test_mskor_lane_order_property)\n")
        f.write(out.stdout)


# ------------------------------------------------------------------ b-3: block-granular, bounded-memory stream decode
class _TrackedReader:
    """A reader over a bytes object that hands out at most `step` bytes per call and remembers how far it got."""

    def __init__(self, data, step):
        self.data, self.step, self.pos, self.calls = data, step, 0, 0

    def __call__(self, _user, p, cap):
        import ctypes as C
        k = min(cap, self.step, len(self.data) - self.pos)
        if k:
            C.memmove(p, self.data[self.pos:self.pos + k], k)
        self.pos += k
        self.calls += 1
        return k


def _stream_decode(ctx, ffi, fmt, stream, step, out_chunk, flags=0):
    """→ (crc32 of the output, output length, first-read reader position, peak lfx_decoder_buffered, status)"""
    import ctypes as C
    import zlib
    rd = _TrackedReader(stream, step)
    cb = ffi.READ_CB(rd)
    st = C.c_int(0)
    d = ffi.lib().lfx_decoder_new(ctx.handle, fmt, flags, cb, None, C.byref(st))
    assert d, st.value
    buf = (C.c_uint8 * out_chunk)()
    crc, total, first_pos, peak, status = 0, 0, None, 0, 0
    while True:
        k = ffi.lib().lfx_decoder_read(d, buf, out_chunk)
        if first_pos is None:
            first_pos = rd.pos
        peak = max(peak, ffi.lib().lfx_decoder_buffered(d))
        if k <= 0:
            status = -k
            break
        crc = zlib.crc32(memoryview(buf)[:k], crc)
        total += k
    consumed = ffi.lib().lfx_decoder_consumed(d)
    ffi.lib().lfx_decoder_free(d)
    return crc, total, first_pos, peak, status, consumed


def test_stream_decoder_bounded_memory_2gib(env):
    """A 2 GiB member (gzip::Encoder, 8192-byte writes: 2048 blocks) through the io::Read-shaped decoder: the decoder
    works a window of blocks at a time (src/deflate/decode.rs:136-164 decodes a block per read and keeps 32 KiB,
    libflate_lz77/src/lib.rs:219-231) — the first read returns long before the reader is drained, and what it buffers
    never exceeds 256 MiB."""
    import ctypes as C
    import zlib
    import torch
    lfx, ctx, ffi, synth = env
    n = 2 << 30
    piece = synth.text(256 << 20)                      # 8 x the same 256 MiB: the encoder's chunks never reach across 256 KiB
    d_in = torch.from_numpy(piece).cuda().repeat(8)
    opts, sched = ffi.make_opts(mtime=0), ffi.make_schedule(8192)
    bound = ffi.lib().lfx_encode_bound(n, C.byref(opts), C.byref(sched)) & ~3
    d_out = torch.empty(bound, dtype=torch.uint8, device="cuda")
    m = ctx.encode_device(ffi.GZIP, d_in.data_ptr(), n, d_out.data_ptr(), bound, opts, sched)
    member = d_out[:m].cpu().numpy().tobytes()
    del d_out
    want_crc = 0
    pb = piece.tobytes()
    for _ in range(8):
        want_crc = zlib.crc32(pb, want_crc)
    assert int.from_bytes(member[-8:-4], "little") == want_crc
    del d_in
    torch.cuda.empty_cache()
    crc, total, first_pos, peak, status, consumed = _stream_decode(ctx, ffi, ffi.GZIP, member, 4 << 20, 16 << 20)
    assert status == 0 and total == n and crc == want_crc and consumed == len(member)
    assert first_pos < len(member) // 8, (first_pos, len(member))          # the first bytes came out of the first windows
    assert peak <= (256 << 20), peak
    print("stream decode of a 2 GiB member: %d compressed bytes, first read after %d, peak buffered %d MiB" % (
        len(member), first_pos, peak >> 20))


def test_stream_decoder_windows_vs_oracle(env, oracle):
    """Windows on everything the block-parallel path distinguishes: reference-made members (blocks never read earlier
    blocks), a foreign encoder's (python zlib / gzip: every block reads the 32 KiB in front of it — the marker path, seeded
    with the previous window's tail), a member written by ONE write (one block larger than a window: the window grows),
    multi-member input, a truncated and a corrupted member (same verdict, same bytes in front of it as the one-shot
    decode).  Readers that hand out 1 MiB and 100 000 bytes at a time."""
    import gzip as pygzip
    import zlib
    lfx, ctx, ffi, synth = env
    text = synth.text(120 << 20).tobytes()
    low = synth.lowent(200 << 20).tobytes()
    ref = ctx.encode_host(ffi.GZIP, text, ffi.make_opts(mtime=0), ffi.make_schedule(8192))          # ~58 MB: 4 windows
    one = ctx.encode_host(ffi.ZLIB, text[:(40 << 20)], ffi.make_opts(), ffi.make_schedule(0))       # one ~19 MB block
    foreign = pygzip.compress(text, 6, mtime=0)                                                      # ~45 MB
    lows = ctx.encode_host(ffi.ZLIB, low, ffi.make_opts(), ffi.make_schedule(8192))                  # 6 MB → 200 MB out
    cases = [("reference gzip", ffi.GZIP, ref, text), ("one block", ffi.ZLIB, one, text[:(40 << 20)]),
             ("python gzip", ffi.GZIP, foreign, text), ("lowent zlib", ffi.ZLIB, lows, low)]
    for name, fmt, stream, plain in cases:
        for step in (1 << 20, 100000):
            crc, total, first_pos, peak, status, consumed = _stream_decode(ctx, ffi, fmt, stream, step, 8 << 20)
            assert status == 0 and total == len(plain) and crc == zlib.crc32(plain) and consumed == len(stream), (name, step, status, total)
            assert peak <= (256 << 20), (name, peak)
        if len(stream) > (40 << 20):
            assert first_pos < len(stream) // 2, (name, first_pos)
    # two members, MultiDecoder (gzip.rs:1142-1166)
    two = ref + foreign
    crc, total, _, _, status, consumed = _stream_decode(ctx, ffi, ffi.GZIP, two, 1 << 20, 8 << 20, flags=ffi.DEC_MULTI)
    assert status == 0 and total == 2 * len(text) and consumed == len(two) and crc == zlib.crc32(text + text)
    # a truncated member and a corrupted one: verdict and delivered bytes as the one-shot decode / the oracle give them
    cut = ref[:len(ref) * 3 // 5]
    bad = bytearray(ref); bad[len(ref) * 3 // 5] ^= 0x55; bad = bytes(bad)
    for name, s in (("truncated", cut), ("corrupted", bad)):
        want = oracle.decode(oracle.GZIP, s)
        crc, total, _, _, status, _ = _stream_decode(ctx, ffi, ffi.GZIP, s, 1 << 20, 8 << 20)
        assert status == {1: ffi.E_INVALID_DATA, 2: ffi.E_UNEXPECTED_EOF}[want[0]], (name, status, want[0], want[3])
        # the reference hands out the bytes of complete blocks before the error (decode.rs:136-164)
        assert total <= len(want[1]) and crc == zlib.crc32(want[1][:total]) and len(want[1]) - total < (2 << 20), (name, total, len(want[1]))


# ------------------------------------------------------------------ e: N-GPU decode of one member, no encoder layout
def _virtual_rank_decode(lfx, ffi, member, hdr_len, world, plain_len):
    """`world` virtual ranks on one device (a context each: a rank's scan tables live in its context): every rank sees only
    its byte range plus the tail, the tuples are concatenated in rank order (what the all-gather yields)."""
    import ctypes as C
    import torch
    from libflate_amd import sharded
    d_member = torch.frombuffer(bytearray(member), dtype=torch.uint8).cuda()
    ranges = sharded.byte_ranges(hdr_len, len(member), world)
    ctxs = [lfx.Context(0) for _ in range(world)]
    # the finder's tail rule (BFINAL headers only near the member's end), and the retry without it when the chain breaks
    ffb = sharded.final_from(len(member))
    while True:
        parts, rows, total_cnt = [], [], 0
        for r, (lo, hi) in enumerate(ranges):
            n_part = min(hi + sharded.RANGE_TAIL, len(member)) - lo
            d_part = d_member[lo:lo + n_part].clone()                     # (its own buffer: nothing outside it can be read)
            tuples, cnt = sharded.range_scan(ctxs[r], r, d_part.data_ptr(), n_part, lo, hi, hdr_len * 8 if r == 0 else None,
                                             final_from_bit=ffb)
            parts.append((d_part, n_part, lo))
            rows.append((tuples, cnt))
            total_cnt += cnt
        tsz = C.sizeof(ffi.BlkTuple)
        all_t = (ffi.BlkTuple * max(total_cnt, 1))()
        at = 0
        for tuples, cnt in rows:
            C.memmove(C.byref(all_t, at * tsz), tuples, cnt * tsz)
            at += cnt
        try:
            chain, nch, total = sharded.chain_of(all_t, total_cnt, hdr_len * 8)
            break
        except ffi.LfxError:
            if not ffb:
                raise
            ffb = 0
    assert total == plain_len
    out = torch.zeros(plain_len, dtype=torch.uint8, device="cuda")
    checks, owned, slices, states = [], [], [], []
    for r, (d_part, n_part, lo) in enumerate(parts):
        d_slice = torch.zeros(plain_len, dtype=torch.uint8, device="cuda")
        ol, base, state = sharded.range_emit(ctxs[r], r, d_part.data_ptr(), n_part, lo, all_t, chain, nch, d_slice.data_ptr(), plain_len)
        slices.append((d_slice, ol, base))
        states.append(state)
        owned.append(ol)
    d_maps = None
    if any(states):           # window hand-over (round 4): the ranks' index maps, "all-gathered" = stacked in rank order
        d_maps = torch.empty((world, 32768), dtype=torch.int16, device="cuda")
        for r in range(world):
            sharded.range_map(ctxs[r], d_maps[r].data_ptr())
    for r in range(world):
        crc, ad = sharded.range_finish(ctxs[r], r, d_maps.data_ptr() if d_maps is not None else None)
        d_slice, ol, base = slices[r]
        out[base:base + ol] = d_slice[:ol]
        checks.append((ol, crc, ad))
    for c in ctxs:
        c.close()
    return out, sharded.fold_checks(checks), owned, nch, total_cnt


def test_member_decode_on_virtual_ranks(env, oracle):
    """north_star: "independent DEFLATE blocks … partition across the GPUs".  An oracle-made 64 MiB gzip member (64 blocks)
    cut into 4 (and 3, 8) byte ranges: every rank finds and scans the blocks that start in its range, the chain is walked
    over the gathered tuples, every rank materialises its blocks — no bit offset comes from the encoder.  Output ==
    input, folded CRC-32 == the trailer's, every rank owns a share."""
    import zlib
    import torch
    lfx, ctx, ffi, synth = env
    data = synth.text(64 << 20)
    plain = data.tobytes()
    member = oracle.encode(oracle.GZIP, plain, write_size=8192, mtime=0)
    for world in (4, 3, 8):
        out, (crc, ad), owned, nch, ncand = _virtual_rank_decode(lfx, ffi, member, 10, world, len(plain))
        assert torch.equal(out, torch.from_numpy(data).cuda()), world
        assert crc == zlib.crc32(plain) == int.from_bytes(member[-8:-4], "little") and ad == zlib.adler32(plain)
        assert nch == 65 and min(owned) > (len(plain) // world) // 2, (world, nch, owned)       # 64 blocks + the empty final one
        assert ncand >= nch
    # a member whose blocks read earlier blocks (python zlib), also across rank boundaries: the window hand-over of round 4
    # (tests/test_gpu_round4.py::test_foreign_member_on_virtual_ranks has the sweep)
    import gzip as pygzip
    foreign = pygzip.compress(plain[:(16 << 20)], 6, mtime=0)
    out, (crc, ad), owned, nch, ncand = _virtual_rank_decode(lfx, ffi, foreign, 10, 4, 16 << 20)
    assert torch.equal(out, torch.from_numpy(data[:16 << 20]).cuda()) and crc == zlib.crc32(plain[:16 << 20])


def test_gzip_header_empty_name_and_comment(env, oracle):
    """A present-but-empty FNAME / FCOMMENT is Some("") in the reference (gzip.rs:415-431), not None."""
    lfx, ctx, ffi, synth = env
    data = kat.test_i()[:5000]
    s = oracle.encode(oracle.GZIP, data, 0, filename=b"", comment=b"", mtime=9)
    assert s[3] & 0x18 == 0x18                                   # FNAME and FCOMMENT present
    d = lfx.gzip.Decoder.new(s)
    h = d.header()
    assert h["filename"] == b"" and h["comment"] == b"" and h["modification_time"] == 9
    assert d.read_to_end() == data
    s2 = oracle.encode(oracle.GZIP, data, 0, mtime=9)
    assert lfx.gzip.Decoder.new(s2).header()["filename"] is None


def test_final_block_outside_the_finder_tail(env, oracle, monkeypatch):
    """The block finder reports headers with BFINAL set only near the end of the stream (lfx_decode.cpp, final_from);
    a last block that starts earlier is scanned on demand by the chain walk.  LFX_NO_FINAL_CAND drops every BFINAL
    candidate, so the last block of every multi-block member takes that path (decode.rs:112-164: same bytes)."""
    lfx, ctx, ffi, synth = env
    monkeypatch.setenv("LFX_NO_FINAL_CAND", "1")
    c1 = lfx.Context(0)
    monkeypatch.delenv("LFX_NO_FINAL_CAND")
    try:
        for n in (3 * (1 << 20) + 12345, (12 << 20) + 7):           # the last block: a remainder of ordinary size
            data = synth.text(n).tobytes()
            s = oracle.encode(oracle.GZIP, data, write_size=8192, mtime=0)
            rc, out, used, msg = c1.decode_host(ffi.GZIP, s, n)
            assert rc == 0 and used == len(s) and out == data, (n, rc, msg)
            rc, out, used, msg = ctx.decode_host(ffi.GZIP, s, n)
            assert rc == 0 and used == len(s) and out == data, (n, rc, msg)
    finally:
        c1.close()
