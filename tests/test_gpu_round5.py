"""GPU: round-5 parity cases — the candidate stage's two pipeline instances (lfx_match7.hip: one that short-cuts repeated
prefixes, picked per segment from a 4 KiB sample of its first bytes; one that knows nothing of them).  The sample is a guess:
either instance has to give DefaultLz77Encoder's answers (libflate_lz77 default.rs:70-131) on any data, so the cases here
put runs where the sample does not see them and text where it promised runs, across segment borders and chunk borders."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_gpu_parity import ctx, enc, ffi, lfx, synth  # noqa: F401  (fixtures)


def pieces(synth):
    rng = np.random.default_rng(5)
    text = synth.text(1 << 20).tobytes()
    return dict(
        text=text, zeros=bytes(1 << 20), rnd=rng.integers(0, 256, 1 << 18, dtype=np.uint8).tobytes(),
        lowent=synth.lowent(1 << 20).tobytes(),
        # runs of every short length between stretches of text: the marks at a run's first / last position
        shortruns=b"".join(text[i * 37:i * 37 + 1 + i % 29] + bytes([65 + i % 7]) * (1 + i % 11) for i in range(20000)),
        # period-2 and period-3 repeats: equal prefixes two and three apart, none adjacent
        period2=b"ab" * 100000, period3=b"abc" * 70000,
    )


def test_runs_and_text_in_one_stream(ctx, ffi, oracle, synth):
    p = pieces(synth)
    cases = {
        "sample sees text, runs behind": p["text"][:70000] + p["zeros"][:400000] + p["text"][:100000],
        "sample sees runs, text behind": p["zeros"][:5000] + p["text"] + p["zeros"][:3],
        "alternating 4 KiB": b"".join((p["zeros"] if i & 1 else p["text"])[i * 4096:(i + 1) * 4096] for i in range(200)),
        "alternating 64 KiB": b"".join((p["lowent"] if i & 1 else p["text"])[i * 65536:(i + 1) * 65536] for i in range(16)),
        "short runs": p["shortruns"], "period 2": p["period2"], "period 3": p["period3"],
        "random then runs": p["rnd"] + bytes([7]) * 300000 + p["rnd"][:1000],
        "run to the last byte": p["text"][:33000] + b"x" * 258 * 40,
        "one run, 2 MiB": b"\xff" * (2 << 20),
    }
    for name, data in cases.items():
        for ws in (0, 8192, 100000):
            got = enc(ctx, ffi, ffi.ZLIB, data, ws)
            assert got == oracle.encode(ffi.ZLIB, data, write_size=ws), (name, ws, len(got))
            st, out, used, _ = ctx.decode_host(ffi.ZLIB, got)
            assert (st, used) == (0, len(got)) and out == data, (name, ws)


def test_runs_with_small_windows_and_lengths(ctx, ffi, oracle, synth):
    """window_size / max_length options (lz77 default.rs:33-57) with runs: the short-cut distance 1 has to stay inside a
    256-byte window too, and a run longer than max_length restarts."""
    p = pieces(synth)
    data = p["text"][:50000] + p["zeros"][:200000] + p["shortruns"][:200000]
    for kw in (dict(window_size=256), dict(window_size=1024, max_length=16), dict(max_length=3), dict(max_length=258)):
        got = enc(ctx, ffi, ffi.GZIP, data, 8192, **kw)
        assert got == oracle.encode(ffi.GZIP, data, write_size=8192, **kw), kw


def test_long_matches_at_short_distances_decode(ctx, ffi, oracle, synth):
    """K3's tiles of long matches (materialize2_body, PERIODIC): a match that overlaps itself — rle_decode's forward copy,
    libflate_lz77 lib.rs:186-190 — takes its bytes from its first period.  Every distance 1..40 and a few up to 300 against
    lengths up to 258, runs that cross tiles and units, single block (write_size 0) and reference-made chunks; members of
    another encoder's making (zlib level 9: lazy matches, distances the default encoder does not choose) as well."""
    import zlib
    rng = np.random.default_rng(11)
    parts = []
    for d in list(range(1, 41)) + [63, 64, 65, 127, 128, 129, 255, 256, 257, 258, 259, 300]:
        seed = rng.integers(0, 256, d, dtype=np.uint8).tobytes()
        for reps in (258 // d + 1, 3000 // d + 2):
            parts.append(seed * reps + rng.integers(0, 256, 5, dtype=np.uint8).tobytes())
    data = b"".join(parts) * 3
    for ws in (0, 8192):
        got = enc(ctx, ffi, ffi.ZLIB, data, ws)
        assert got == oracle.encode(ffi.ZLIB, data, write_size=ws), ws
        st, out, used, _ = ctx.decode_host(ffi.ZLIB, got)
        assert (st, used) == (0, len(got)) and out == data, ws
    for level in (1, 6, 9):
        z = zlib.compress(data, level)
        st, out, used, _ = ctx.decode_host(ffi.ZLIB, z)
        assert (st, used) == (0, len(z)) and out == data, level
    lo = synth.lowent(8 << 20).tobytes()
    z = zlib.compress(lo, 9)
    st, out, used, _ = ctx.decode_host(ffi.ZLIB, z)
    assert (st, used) == (0, len(z)) and out == lo


def test_block_headers_of_many_shapes_decode(ctx, ffi, synth):
    """The code-length sequence of a dynamic block header (decode.rs:166-223 / symbol.rs:245-331: symbols 0-15 literal widths, 16
    repeat the previous width 3-6 times, 17 / 18 runs of zeros) is decoded 64 bit offsets at a time (parse_header, round 5): a
    repeat whose previous width lies in the window before, runs of zeros across the literal / distance boundary, alphabets of
    two symbols and of 286.  Streams of another encoder (python-zlib: every strategy, small memLevel = many blocks), single
    stream and batch path, against the bytes that went in."""
    import zlib
    rng = np.random.default_rng(23)
    text = synth.text(1 << 19).tobytes()
    inputs = {
        "text": text,
        "two symbols": rng.integers(0, 2, 200000, dtype=np.uint8).tobytes(),
        "sparse alphabet": bytes(rng.choice(np.array([0, 7, 64, 200, 255], dtype=np.uint8), 200000)),
        "all byte values, flat": rng.integers(0, 256, 150000, dtype=np.uint8).tobytes(),
        "all byte values, skewed": bytes(np.minimum(rng.geometric(0.03, 200000), 255).astype(np.uint8)),
        "ramp": bytes(range(256)) * 400,
        "text + zeros": text[:100000] + bytes(50000) + text[100000:150000],
    }
    streams = []
    for name, data in inputs.items():
        for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
            for mem in (1, 8):
                for level in (1, 6, 9):
                    co = zlib.compressobj(level, zlib.DEFLATED, 15, mem, strategy)
                    streams.append((name, strategy, mem, level, co.compress(data) + co.flush(), data))
    for name, strategy, mem, level, z, data in streams:
        st, out, used, msg = ctx.decode_host(ffi.ZLIB, z)
        assert (st, used) == (0, len(z)) and out == data, (name, strategy, mem, level, msg)
    # the batch path (blocks of many streams per launch): every fourth stream
    import torch
    sub = streams[::4]
    zs, ds = [t[4] for t in sub], [t[5] for t in sub]
    in_len = np.array([len(z) for z in zs], dtype=np.uint64)
    in_off = (np.cumsum(in_len) - in_len).astype(np.uint64)
    out_cap = np.array([len(d) for d in ds], dtype=np.uint64)
    out_off = (np.cumsum(out_cap) - out_cap).astype(np.uint64)
    d_in = torch.frombuffer(bytearray(b"".join(zs)), dtype=torch.uint8).cuda()
    d_out = torch.zeros(int(out_cap.sum()), dtype=torch.uint8, device="cuda")
    out_len = np.zeros(len(sub), dtype=np.uint64)
    status = np.zeros(len(sub), dtype=np.int32)
    torch.cuda.synchronize()
    rc = ffi.lib().lfx_decode_batch_device(ctx.handle, ffi.ZLIB, len(sub), d_in.data_ptr(), in_off.ctypes.data, in_len.ctypes.data,
                                           d_out.data_ptr(), out_off.ctypes.data, out_cap.ctypes.data, out_len.ctypes.data,
                                           status.ctypes.data)
    assert rc == 0 and not status.any(), (rc, status.nonzero())
    host = d_out.cpu().numpy().tobytes()
    for i, (name, strategy, mem, level, z, data) in enumerate(sub):
        assert host[int(out_off[i]):int(out_off[i]) + int(out_len[i])] == data, (name, strategy, mem, level)
