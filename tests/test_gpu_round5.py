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
