"""CPU: the host planner and the host/device-shared Huffman + header code (lfx_plan.h, lfx_huff.h —
the SAME source the HIP kernels compile) reproduce the oracle bit for bit when the GPU-only stages
(match, parse, pack) are stood in for by straightforward Python."""
import ctypes as C
import os

import numpy as np
import pytest

from golden import kat


@pytest.fixture(scope="module")
def ffi():
    import __graft_entry__ as g
    g.build()
    from libflate_amd import _ffi
    return _ffi


def huff_block(ffi, hist320, btype):
    hist = np.ascontiguousarray(hist320, dtype=np.uint32)
    lit = np.zeros(288, np.uint32); dist = np.zeros(32, np.uint32); hdr = np.zeros(160, np.uint32)
    hb, bb = C.c_uint32(0), C.c_uint64(0)
    ffi.lib().lfx_debug_huff_block(hist.ctypes.data, btype, lit.ctypes.data, dist.ctypes.data,
                                   hdr.ctypes.data, C.byref(hb), C.byref(bb))
    return lit, dist, hdr, hb.value, bb.value


def plan(ffi, fmt, n, write_size=0, writes=None, **kw):
    o = ffi.make_opts(**kw)
    s = ffi.make_schedule(write_size, writes)
    ch = np.zeros(4 * 4096, np.uint64); bl = np.zeros(6 * 4096, np.uint64)
    nc, nb = C.c_size_t(0), C.c_size_t(0)
    rc = ffi.lib().lfx_debug_plan(fmt, C.byref(o), C.byref(s), n, ch.ctypes.data, 4096, C.byref(nc),
                                  bl.ctypes.data, 4096, C.byref(nb))
    assert rc == 0
    return ch[:4 * nc.value].reshape(-1, 4), bl[:6 * nb.value].reshape(-1, 6)


def symbols(ffi, length, distance):
    out = np.zeros(6, np.uint32)
    ffi.lib().lfx_debug_symbols(length, distance, out.ctypes.data)
    return [int(x) for x in out]


class Bits:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, width, bits):
        self.acc |= bits << self.n
        self.n += width

    def align(self):
        self.n = (self.n + 7) // 8 * 8

    def bytes(self):
        return self.acc.to_bytes((self.n + 7) // 8, "little")


def emulate_deflate(ffi, oracle, data, write_size=0, writes=None, **kw):
    """raw DEFLATE bytes assembled from: planner (C++) + oracle LZ77 per chunk + lfx_huff.h (C++)."""
    chunks, blocks = plan(ffi, ffi.DEFLATE, len(data), write_size, writes, **kw)
    bw = Bits()
    window, maxlen = kw.get("window_size", 32768), kw.get("max_length", 258)
    for btype, final, first, nch, in_off, in_len in blocks:
        btype, final = int(btype), int(final)
        bw.put(1, final); bw.put(2, btype)
        if btype == 0:
            bw.align()
            ln = int(in_len)
            bw.put(16, ln); bw.put(16, (~ln) & 0xFFFF)
            for b in data[int(in_off):int(in_off) + ln]:
                bw.put(8, b)
            continue
        codes = []
        for c in chunks[int(first):int(first) + int(nch)]:
            off, ln, _blk, flags = (int(x) for x in c)
            seg = data[off:off + ln]
            if flags & 2:
                codes.extend(int(b) << 16 for b in seg)
            else:
                codes.extend(int(x) for x in oracle.lz77_chunk(seg, window, maxlen))
        codes.append(256 << 16)
        hist = np.zeros(320, np.uint32)
        for w in codes:
            val, dist = w >> 16, w & 0xFFFF
            if dist == 0:
                hist[val] += 1
            else:
                s = symbols(ffi, val, dist)
                hist[s[0]] += 1
                hist[288 + s[3]] += 1
        lit, dst, hdr, hbits, body = huff_block(ffi, hist, btype)
        start = bw.n
        for i in range(hbits):
            bw.put(1, (int(hdr[i >> 5]) >> (i & 31)) & 1)
        for w in codes:
            val, dist = w >> 16, w & 0xFFFF
            if dist == 0:
                e = int(lit[val]); bw.put(e >> 16, e & 0xFFFF)
            else:
                s = symbols(ffi, val, dist)
                e = int(lit[s[0]]); bw.put(e >> 16, e & 0xFFFF)
                if s[1]: bw.put(s[1], s[2])
                e = int(dst[s[3]]); bw.put(e >> 16, e & 0xFFFF)
                if s[4]: bw.put(s[4], s[5])
        assert bw.n - start + 3 == body, "body_bits mismatch"
    bw.align()
    return bw.bytes()


def test_symbol_maps(ffi):
    # closed forms == Symbol::code/extra_lengh/distance (symbol.rs:95-154) via the oracle's tables
    LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115,
                131, 163, 195, 227, 258]
    LEN_EXTRA = [0] * 8 + [1] * 4 + [2] * 4 + [3] * 4 + [4] * 4 + [5] * 4 + [0]
    DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537,
                 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
    DIST_EXTRA = [0, 0, 0, 0] + [i // 2 for i in range(2, 28)]
    for length in range(3, 259):
        s = symbols(ffi, length, 1)
        k = s[0] - 257
        assert 0 <= k < 29 and s[1] == LEN_EXTRA[k] and LEN_BASE[k] + s[2] == length
        if length == 258:
            assert s[0] == 285
        else:
            assert k < 28 and s[2] < (1 << LEN_EXTRA[k])
    for d in list(range(1, 600)) + [1024, 1025, 4096, 8193, 16384, 16385, 24576, 24577, 32767, 32768]:
        s = symbols(ffi, 3, d)
        assert DIST_BASE[s[3]] + s[5] == d and s[4] == DIST_EXTRA[s[3]] and s[5] < (1 << s[4] if s[4] else 1)


def test_huffman_vs_oracle(ffi, oracle):
    rng = np.random.default_rng(11)
    cases = []
    for trial in range(120):
        hist = np.zeros(320, np.uint32)
        kind = trial % 6
        nlit = int(rng.integers(1, 286))
        if kind == 0:
            hist[:286] = rng.integers(0, 1000, 286)
        elif kind == 1:   # skewed: forces the length-limiting branch
            v = np.array([int(1.6 ** i) for i in range(40)], dtype=np.uint64) % (1 << 31)
            idx = rng.permutation(286)[:40]
            hist[idx] = v.astype(np.uint32)
        elif kind == 2:   # many ties
            hist[rng.permutation(286)[:nlit]] = rng.integers(1, 4, nlit)
        elif kind == 3:   # sparse
            hist[rng.permutation(286)[:rng.integers(1, 5)]] = rng.integers(1, 100000, 1)
        elif kind == 4:
            hist[:286] = (rng.pareto(0.7, 286) * 10).astype(np.uint32)
        else:
            hist[:286] = rng.integers(0, 2, 286) * rng.integers(1, 1 << 20, 286)
        hist[256] = 1
        nd = int(rng.integers(0, 31))
        if nd and kind != 3:
            hist[288 + rng.permutation(30)[:nd]] = rng.integers(1, 5000, nd) if kind != 1 else \
                np.array([int(1.9 ** i) for i in range(nd)], dtype=np.uint32)
        cases.append(hist)
    # (round 4) frequencies of 2^23 and more take the rank sort's pair compare instead of the single 32-bit key; Fibonacci
    # weights and powers of two make the deepest trees and the longest runs of equal weights in the depth merge's queue
    for trial in range(60):
        hist = np.zeros(320, np.uint32)
        nlit = int(rng.integers(2, 286))
        idx = rng.permutation(286)[:nlit]
        kind = trial % 5
        if kind == 0:
            hist[idx] = rng.integers(1 << 22, 1 << 25, nlit)
        elif kind == 1:
            hist[idx] = rng.integers(1, 3, nlit)
            hist[idx[0]] = 1 << 24
        elif kind == 2:
            v = [1, 1]
            while len(v) < nlit:
                v.append((v[-1] + v[-2]) % (1 << 31) or 1)
            hist[idx] = np.array(v[:nlit], dtype=np.uint64)
        elif kind == 3:
            hist[idx] = 2 ** rng.integers(0, 6, nlit)
        else:
            hist[idx] = np.minimum(rng.geometric(0.3, nlit), 255)
        hist[256] = max(1, int(hist[256]))
        nd = int(rng.integers(0, 31))
        if nd:
            hist[288 + rng.permutation(30)[:nd]] = rng.integers(1, 4, nd) if kind >= 3 else rng.integers(1, 1 << 26, nd)
        cases.append(hist)
    e = np.zeros(320, np.uint32); e[256] = 1
    cases.append(e)   # the empty final block
    for hist in cases:
        lit, dst, hdr, hbits, body = huff_block(ffi, hist, 2)
        lw = oracle.huff_widths(hist[:286], 15)
        dh = hist[288:318].copy()
        if dh.sum() == 0:
            dh[0] = 1                                     # symbol.rs:332-337
        dw = oracle.huff_widths(dh, 15)
        assert list(lit[:286] >> 16) == list(lw)
        assert list(dst[:30] >> 16) == list(dw)
        assert list(lit[:286] & 0xFFFF) == list(oracle.huff_codes(lw))
        assert list(dst[:30] & 0xFFFF) == list(oracle.huff_codes(dw))


class _CannedCodes:
    """An `E: Lz77Encode` that ignores its input and hands out a prepared code list on the first flush."""
    def __init__(self, codes):
        self.codes = list(codes)

    def encode(self, buf, sink):
        pass

    def flush(self, sink):
        sink.extend(self.codes)
        self.codes = []

    def compression_level(self):
        return 2

    def window_size(self):
        return 32768


def _one_final_block(ffi, codes):
    """raw DEFLATE of ONE final dynamic block holding `codes`, header and code tables from lfx_huff.h"""
    words = [(c[1] << 16) if c[0] == "Literal" else (c[1] << 16) | c[2] for c in codes] + [256 << 16]
    hist = np.zeros(320, np.uint32)
    syms = []
    for w in words:
        val, dist = w >> 16, w & 0xFFFF
        if dist == 0:
            hist[val] += 1
            syms.append(None)
        else:
            s = symbols(ffi, val, dist)
            hist[s[0]] += 1
            hist[288 + s[3]] += 1
            syms.append(s)
    lit, dst, hdr, hbits, body = huff_block(ffi, hist, 2)
    bw = Bits()
    bw.put(1, 1); bw.put(2, 2)
    for i in range(hbits):
        bw.put(1, (int(hdr[i >> 5]) >> (i & 31)) & 1)
    for w, s in zip(words, syms):
        if s is None:
            e = int(lit[w >> 16]); bw.put(e >> 16, e & 0xFFFF)
        else:
            e = int(lit[s[0]]); bw.put(e >> 16, e & 0xFFFF)
            if s[1]: bw.put(s[1], s[2])
            e = int(dst[s[3]]); bw.put(e >> 16, e & 0xFFFF)
            if s[4]: bw.put(s[4], s[5])
    assert bw.n == body
    bw.align()
    return bw.bytes()


def test_block_headers_of_crafted_alphabets(ffi, oracle):
    """DynamicHuffmanCodec::save / build_bitwidth_codes (symbol.rs:343-386,486-540) on alphabets made to hit every branch
    of the run-length code: zero runs of 1..2, 3..10, 11..138, 139, 148, 149, 276 and more, runs of equal widths of every
    length around the groups of six, a run that ends where the distance table starts, one-symbol and empty tables.  The
    reference side is the oracle's encoder fed the same code words through a caller-made `E: Lz77Encode`."""
    rng = np.random.default_rng(23)
    LENS = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
    DISTS = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097,
             6145, 8193, 12289, 16385, 24577]
    cases = []
    # literal i used iff it lies in one of the ranges; every used literal once → long runs of one width
    def lits(ranges, rep=1):
        out = []
        for lo, hi in ranges:
            for v in range(lo, hi):
                out.extend([("Literal", v)] * rep)
        return out
    for gap in (1, 2, 3, 4, 10, 11, 12, 137, 138, 139, 140, 148, 149, 150, 200, 254):
        cases.append(lits([(0, 1), (gap + 1, min(gap + 3, 256))]))           # a zero run of `gap` between used literals
    for run in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 64, 65, 128, 255, 256):
        cases.append(lits([(0, run)]) + [("Literal", 0)] * 3)                 # `run` literals, equal widths for most
    cases.append(lits([(0, 256)]))                                            # 256 equal widths + EOB
    cases.append([])                                                          # the empty block: EOB only, dist[0] dummy
    cases.append([("Pointer", 258, 32768)] * 5 + [("Literal", 255)])
    cases.append([("Pointer", LENS[k], DISTS[k]) for k in range(29)] + [("Pointer", 3, DISTS[29])])   # every length / distance symbol once
    cases.append(lits([(0, 256)]) + [("Pointer", LENS[k], DISTS[k]) for k in range(29)] + [("Pointer", 3, DISTS[29])])
    cases.append(lits([(250, 256)]) + [("Pointer", 258, 1)])                  # widths up to symbol 285, one distance
    cases.append(lits([(250, 256)], rep=2) + [("Pointer", 258, 24577)] * 2)   # ... the last distance symbol: zero run of 29 in the distance table
    for trial in range(60):
        codes = []
        k = int(rng.integers(1, 6))
        for _ in range(k):
            lo = int(rng.integers(0, 256)); hi = min(256, lo + int(rng.integers(1, 80)))
            codes += lits([(lo, hi)], rep=int(rng.integers(1, 4)))
        for _ in range(int(rng.integers(0, 40))):
            codes.append(("Pointer", int(rng.choice(LENS)) if rng.random() < 0.7 else int(rng.integers(3, 259)),
                          int(rng.choice(DISTS)) if rng.random() < 0.7 else int(rng.integers(1, 32769))))
        rng.shuffle(codes)
        cases.append([tuple(c) for c in codes])
    for codes in cases:
        codes = [(c[0], int(c[1])) if c[0] == "Literal" else (c[0], int(c[1]), int(c[2])) for c in codes]
        want = oracle.encode(oracle.DEFLATE, b"x", **oracle.custom_lz77(_CannedCodes(codes)))
        got = _one_final_block(ffi, codes)
        assert got == want, codes[:8]


def test_emulated_pipeline_matches_oracle(ffi, oracle):
    rng = np.random.default_rng(5)
    text = kat.test_i()
    cases = [
        (kat.HELLO, dict()), (b"", dict()), (b"a", dict()), (b"aaaaa", dict()),
        (b"hello hello hello", dict()),
        (kat.ISSUE52[:16031], dict()), (kat.ISSUE52, dict()),
        (text, dict(write_size=8192)), (text, dict(write_size=1000, block_size=20000)),
        (text, dict(dynamic_huffman=0)), (text[:5000], dict(lz77_kind=1)),
        (text[:70000 * 2], dict(no_compression=1)), (text[:300], dict(no_compression=1, block_size=100, write_size=70)),
        (text, dict(window_size=1024, max_length=16)),
        (bytes(300000), dict()), (bytes(300000), dict(write_size=8192)),
        (rng.integers(0, 256, 40000, dtype=np.uint8).tobytes(), dict(write_size=8192, block_size=16384)),
        (kat.ramp()[:300000], dict(write_size=8192, block_size=65536)),
    ]
    for data, kw in cases:
        ws = kw.pop("write_size", 0)
        got = emulate_deflate(ffi, oracle, data, ws, None, **kw)
        want = oracle.encode(oracle.DEFLATE, data, write_size=ws, **kw)
        assert got == want, (len(data), kw, ws)
    assert emulate_deflate(ffi, oracle, kat.HELLO) == kat.DEFLATE_HELLO   # encode.rs:152-154
    # flush events (zlib.rs:840-902 shapes, raw deflate here)
    writes = [18, 3, 3, None, 18, 3, 3, None]
    got = emulate_deflate(ffi, oracle, kat.ISSUE27_PLAIN, 0, writes)
    assert got == kat.ISSUE27_ZLIB_NONE[2:-4]


def test_structural_insight_parse_independent_candidates(oracle):
    """cand(i) = most recent earlier same-trigram position is parse independent, so the
    greedy walk over per-position (len, dist) answers reproduces DefaultLz77Encoder::flush."""
    rng = np.random.default_rng(9)
    inputs = [kat.test_i()[:20000], bytes(5000), rng.integers(0, 3, 20000, dtype=np.uint8).tobytes(),
              kat.ISSUE52, b"abcabcabcabcabcabc" * 50, b"ab", b"abc", b"abcd"]
    for window, maxlen in ((32768, 258), (64, 258), (32768, 8), (300, 20)):
        for data in inputs:
            n = len(data)
            end = max(3, n) - 3
            last = {}
            md = []
            for i in range(end):
                key = data[i:i + 3]
                j = last.get(key)
                last[key] = i
                if j is not None and i - j <= window:
                    lim = min(n - (i + 3), maxlen - 3)
                    l = 0
                    while l < lim and data[i + 3 + l] == data[j + 3 + l]:
                        l += 1
                    md.append((3 + l, i - j))
                else:
                    md.append((0, 0))
            codes, i = [], 0
            while i < end:
                ln, d = md[i]
                if d:
                    codes.append((ln << 16) | d); i += ln
                else:
                    codes.append(data[i] << 16); i += 1
            codes.extend(b << 16 for b in data[i:])
            assert codes == [int(x) for x in oracle.lz77_chunk(data, window, maxlen)], (window, maxlen, n)


def test_lazy_parse_model_matches_oracle(oracle):
    """The round-3 parse (lfx_parse2.hip) computes match lengths only at visited positions and rebuilds code words from
    visit bits.  tools/parse2_model.py restates its four kernels statement for statement (lane loops instead of
    wavefronts): speculative group walks, the in-wavefront entry chain with serial repair and jumped-over groups, the
    segment chain, the rebuild in front of a merge point.  On the CPU it must reproduce DefaultLz77Encoder::flush
    (default.rs:69-109) — on text (no repairs), on runs / periodic data (a repair per jumped-over stretch, segments that
    never merge) and at every group / segment boundary size."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import parse2_model as pm
    import synth
    rng = np.random.default_rng(3)
    cases = [(synth.text(30000).tobytes(), {}), (synth.lowent(40000).tobytes(), {}), (bytes(15000), {}),
             (rng.integers(0, 256, 9000, dtype=np.uint8).tobytes(), {}), (b"abc" * 5000, {}), (b"abcdefg" * 2500, {}),
             (rng.integers(0, 4, 20000, dtype=np.uint8).tobytes(), {}),
             (synth.text(20000).tobytes(), {"window": 1024, "max_len": 20}),
             (synth.lowent(15000).tobytes(), {"max_len": 3}), (kat.ISSUE52, {})]
    text = synth.text(7000).tobytes()
    for n in (0, 1, 2, 3, 4, 5, 51, 52, 53, 55, 56, 3327, 3328, 3329, 3330, 3331, 3332, 6655, 6656, 6659):
        cases.append((text[:n], {}))
    saw_repairs = saw_redo = False
    for data, kw in cases:
        st = {}
        got = pm.parse_chunk(data, stats=st, **kw)
        want = oracle.lz77_chunk(data, kw.get("window", 32768), kw.get("max_len", 258))
        assert len(got) == len(want) and (got == want).all(), (len(data), kw)
        saw_repairs |= st.get("repairs", 0) > 0
        saw_redo |= st.get("redo", 0) > 0
    assert saw_repairs and saw_redo       # the corpus reaches the serial repair and the never-merged segment paths


def test_incremental_planner_equals_one_shot(ffi):
    """The stream encoder takes closed blocks out of its planner while later writes are still arriving
    (Planner::take_closed); stitched together, the pieces must be the plan of the whole event list."""
    rng = np.random.default_rng(11)
    for trial in range(60):
        kw = {}
        r = trial % 6
        if r == 1: kw["no_compression"] = 1
        if r == 2: kw["block_size"] = int(rng.choice([1000, 65536, 3 << 20]))
        if r == 3: kw["lz77_kind"] = 1
        if r == 4: kw["window_size"] = 1024
        if r == 5: kw["no_compression"] = 1; kw["block_size"] = 700
        sync = 2 if trial % 4 == 0 else 0
        writes, total = [], 0
        for _ in range(int(rng.integers(1, 120))):
            if rng.integers(0, 6) == 0:
                writes.append(None)
            else:
                w = int(rng.choice([0, 1, 100, 8192, 65535, 65536, 262144, 300000, 2 << 20]))
                writes.append(w); total += w
        fmt = ffi.ZLIB
        o = ffi.make_opts(zlib_flush_mode=sync, **kw)
        s = ffi.make_schedule(writes=writes)
        want_c, want_b = plan(ffi, fmt, total, writes=writes, zlib_flush_mode=sync, **kw)
        for every in (1, 3, 17):
            ch = np.zeros(4 * 8192, np.uint64); bl = np.zeros(6 * 8192, np.uint64)
            nc, nb = C.c_size_t(0), C.c_size_t(0)
            rc = ffi.lib().lfx_debug_plan_incremental(fmt, C.byref(o), C.byref(s), total, every, ch.ctypes.data, 8192,
                                                      C.byref(nc), bl.ctypes.data, 8192, C.byref(nb))
            assert rc == 0
            got_c, got_b = ch[:4 * nc.value].reshape(-1, 4), bl[:6 * nb.value].reshape(-1, 6)
            assert got_c.shape == want_c.shape and (got_c == want_c).all(), (trial, every, kw)
            assert got_b.shape == want_b.shape and (got_b == want_b).all(), (trial, every, kw)


def test_gzip_options_header_carries_the_level(ffi):
    """gzip::EncodeOptions::header (gzip.rs:717-720) replaces the header the options held, compression level included:
    a header cloned from a decoder keeps its XFL (4 Fastest, 2 Slowest, gzip.rs:84-92), a builder's is Unknown."""
    from libflate_amd import gzip
    cloned = dict(modification_time=5, os=3, is_text=False, is_verified=False, extra_field=None, filename=b"a", comment=None)
    for xfl, level in ((4, 2), (2, 4), (0, 3), (7, 3)):
        o = gzip.EncodeOptions().header(dict(cloned, xfl=xfl))
        assert o._kw["lz77_level"] == level and o._kw["mtime"] == 5 and o._kw["filename"] == b"a"
        assert "comment" not in o._kw and "extra" not in o._kw
    o = gzip.EncodeOptions().header(gzip.HeaderBuilder().modification_time(9).finish())
    assert o._kw["lz77_level"] == 3 and o._kw["mtime"] == 9
    # no_compression() resets the level when it is called (gzip.rs:703); header() afterwards replaces it again
    # (gzip.rs:717-720) — call order decides, and the library writes what it is handed
    a = gzip.EncodeOptions().header(dict(cloned, xfl=4)).no_compression()
    b = gzip.EncodeOptions().no_compression().header(dict(cloned, xfl=4))
    assert a._kw["lz77_level"] == 3 and b._kw["lz77_level"] == 2 and a._kw["no_compression"] == b._kw["no_compression"] == 1


def test_header_window_model_equals_serial_walk():
    """parse_header's decode of the code-length sequence by windows of 64 bit offsets (round 5) against the one-symbol-at-a-time
    walk of the reference (decode.rs:166-223, symbol.rs:245-331), as CPU models (tools/hdr_model.py): random complete
    code-length codes, random width sequences with runs (symbols 16 / 17 / 18 across window borders, a 16 whose previous width
    lies one or two windows back), clean and with flipped bits."""
    import random
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import hdr_model as hm
    rng = random.Random(5)
    checked = damaged = 0
    for trial in range(400):
        # a complete code over the 19 code-length symbols: widths from a random binary tree (Kraft sum 1)
        leaves = [0]
        while len(leaves) < rng.randint(2, 19):
            k = rng.randrange(len(leaves))
            if leaves[k] >= 7:
                continue
            d = leaves.pop(k)
            leaves += [d + 1, d + 1]
        syms = rng.sample(range(19), len(leaves))
        cl = [0] * 19
        for s, w in zip(syms, leaves):
            cl[s] = w
        if min(leaves) == 0:
            continue
        tab = hm.build_table(cl)
        usable = [s for s in range(16) if cl[s]]
        if not usable:
            continue
        total = rng.randint(258, 316)
        lengths, style = [], rng.random()
        while len(lengths) < total:
            v = rng.choice(usable)
            lengths += [v] * (rng.randint(1, 40) if style < 0.5 else rng.randint(1, 3))
        lengths = lengths[:total]
        bits = hm.encode_lengths(lengths, cl, rng) + [rng.randint(0, 1) for _ in range(200)]
        a, b = hm.serial(bits, tab, total), hm.windowed(bits, tab, total)
        assert a is not None and a[0] == lengths and a == b, trial
        checked += 1
        for _ in range(3):                       # damage: the two walks have to agree on the verdict and, when clean, on the widths
            hurt = list(bits)
            hurt[rng.randrange(len(bits) - 200)] ^= 1
            a, b = hm.serial(hurt, tab, total), hm.windowed(hurt, tab, total)
            assert a == b, trial
            damaged += a is None
    assert checked > 300 and damaged > 50


def test_periodic_source_quotient_is_exact_or_one_short():
    """K3's tiles of long matches take a self-overlapping match's byte from `offset mod distance` of its first period
    (materialize2_body, PERIODIC; rle_decode, libflate_lz77 lib.rs:186-190), the remainder by a float multiply with v_rcp_f32
    and ONE fix-up: r = off - d * uint(float(off) * rcp(d)); if r >= d: r -= d.  v_rcp_f32 is good to one ulp: for every
    distance and offset that can occur (d <= off < 258) and the reciprocal one ulp low, exact-rounded and one ulp high, the
    truncated quotient is exact or one short — never above, never two short."""
    d = np.arange(1, 258, dtype=np.float32)[:, None]
    off = np.arange(0, 258, dtype=np.float32)[None, :]
    exact = (np.float32(1.0) / d).astype(np.float32)
    for rcp in (np.nextafter(exact, np.float32(0)), exact, np.nextafter(exact, np.float32(2))):
        q = (off * rcp).astype(np.float32).astype(np.int64)           # (float multiply rounded to nearest, then truncated)
        di, oi = d.astype(np.int64), off.astype(np.int64)
        r = oi - di * q
        r = np.where(r >= di, r - di, r)
        live = oi >= di                                               # (the kernel takes this path only for off >= d)
        assert ((r == oi % di) | ~live).all()


def test_bucket_lru_formulation_is_exact(tmp_path):
    """The candidate stage's formulation (lfx_match7.hip): per bucket the most recent entry and the most recent
    entry with ANOTHER tag, bucket + tag = the 24-bit prefix under a bijection, the rest by a walk over duplicate-collapsed
    links — against the exact "most recent earlier occurrence of the three bytes inside the window" (libflate_lz77
    default.rs:76-87), as a CPU model with the kernel's own key multiplier: text, LOWENT, random bytes (55 % of the positions
    go to the walk), a three-letter alphabet, nibbles.  The model exits non-zero on the first wrong answer."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    exe = str(tmp_path / "lru_model")
    subprocess.run([gcc, "-O2", "-o", exe, os.path.join(root, "tools", "exp", "lru_model.c"), os.path.join(root, "tools", "synth.c"), "-lm"],
                   check=True, capture_output=True, timeout=120)
    for kind in range(5):
        r = subprocess.run([exe, str(kind), "2", "6"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "WRONG=0" in r.stdout, (kind, r.stdout, r.stderr)


def test_planner_write_repeat_equals_the_loop_of_writes(tmp_path):
    """Round 6: a fixed-size write schedule (the reference's io::copy protocol: 32768 writes of 8 KiB per 256 MiB) is planned
    by Planner::write_repeat, which takes the writes between two events (an LZ77 flush, a block) in one step — 157 us of host
    time per encode call otherwise.  tests/c/plan_repeat.cpp: the same plan and state as the loop of write() calls over 20000
    random option sets and call sequences."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    assert gxx, "g++ is part of the image"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "plan_repeat")
    subprocess.run([gxx, "-O2", "-std=c++17", "-o", exe, os.path.join(root, "tests", "c", "plan_repeat.cpp")], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "plan_repeat ok: 20000 cases" in out
