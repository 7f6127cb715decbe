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
