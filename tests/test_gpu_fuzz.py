"""GPU: randomized round trips over the size range in which the decoder changes paths (serial kernel, piece walk with and
without warm-up, pieces over candidate ranges, one workgroup per block; direct / marker materialisation; window chain /
blocked prefix; 256- and 1024-lane units) and the encoder changes segmenting.  Every case: our encoder's stream equals the
oracle's byte for byte (DefaultLz77Encoder::flush default.rs:69-109 … Encoder::finish encode.rs:203-249), our decoder
returns the input for our stream AND for a python-zlib stream of the same bytes (foreign block structure, back-references
across blocks: decode.rs:112-164, lib.rs:149-242).

LFX_FUZZ=<n> sets the number of cases (default 120: about half a minute), LFX_FUZZ_SEED the seed, LFX_FUZZ_MINLOG2 / LFX_FUZZ_MAXLOG2 the
size range (default 6 … 23).  The round's soaks: profiles/r04_fuzz.txt."""
import ctypes as C
import os
import sys
import zlib

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


def _data(rng, synth, n, kind):
    if kind == 0:
        return synth.text(n, seed=int(rng.integers(1, 1 << 30))).tobytes()
    if kind == 1:
        return synth.lowent(n, seed=int(rng.integers(1, 1 << 30))).tobytes()
    if kind == 2:
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == 3:      # long runs and long matches
        unit = rng.integers(0, 256, int(rng.integers(1, 700)), dtype=np.uint8).tobytes()
        return (unit * (n // len(unit) + 1))[:n]
    a = synth.text(n, seed=int(rng.integers(1, 1 << 30)))   # text with random stretches
    for _ in range(int(rng.integers(1, 6))):
        lo = int(rng.integers(0, max(n, 1)))
        hi = min(n, lo + int(rng.integers(1, max(n // 4, 2))))
        a[lo:hi] = rng.integers(0, 256, hi - lo, dtype=np.uint8)
    return a.tobytes()


def test_random_round_trips(env, oracle):
    import torch
    lfx, ctx, ffi, synth = env
    cases = int(os.environ.get("LFX_FUZZ", "120"))
    rng = np.random.default_rng(int(os.environ.get("LFX_FUZZ_SEED", "20260927")))
    fmts = ((ffi.GZIP, oracle.GZIP, 31), (ffi.ZLIB, oracle.ZLIB, 15), (ffi.DEFLATE, oracle.DEFLATE, -15))
    lo2, hi2 = float(os.environ.get("LFX_FUZZ_MINLOG2", "6")), float(os.environ.get("LFX_FUZZ_MAXLOG2", "23"))
    for case in range(cases):
        n = int(2 ** rng.uniform(lo2, hi2))                   # 64 B … 8 MiB, log-uniform
        if case % 8 == 7 and hi2 <= 23:
            n = int(rng.choice([4096, 32768, 65536, 262144, 1 << 20, (1 << 20) + 1, 3 << 20]))   # the path boundaries themselves
        data = _data(rng, synth, n, int(rng.integers(0, 5)))
        fmt, ofmt, wbits = fmts[int(rng.integers(0, 3))]
        ws = int(rng.choice([0, 8192, 8192, 1000, 70000]))
        bs = int(rng.choice([1 << 20, 1 << 20, 65536, 300000]))
        kw = dict(mtime=0) if fmt == ffi.GZIP else {}
        opts, sched = ffi.make_opts(block_size=bs, **kw), ffi.make_schedule(ws)
        want = oracle.encode(ofmt, data, write_size=ws, block_size=bs, **kw)
        d_in = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
        bound = ffi.lib().lfx_encode_bound(n, C.byref(opts), C.byref(sched))
        d_out = torch.empty(max(bound, 64), dtype=torch.uint8, device="cuda")
        m = ctx.encode_device(fmt, d_in.data_ptr(), n, d_out.data_ptr(), d_out.numel(), opts, sched)
        got = d_out[:m].cpu().numpy().tobytes()
        assert got == want, ("encode", case, n, ws, bs, fmt)
        d_dec = torch.empty(max(n, 1), dtype=torch.uint8, device="cuda")
        rc, ol, used, msg = ctx.decode_device(fmt, d_out.data_ptr(), m, d_dec.data_ptr(), n)
        assert rc == 0 and ol == n and used == m, ("decode own", case, n, rc, msg)
        assert d_dec[:n].cpu().numpy().tobytes() == data, ("decode own: bytes", case, n)
        # the same bytes from zlib (levels 1 / 6 / 9: different block sizes and match policies)
        co = zlib.compressobj(int(rng.choice([1, 6, 9] if n <= (16 << 20) else [1, 6])), zlib.DEFLATED, wbits)
        foreign = co.compress(data) + co.flush()
        d_f = torch.from_numpy(np.frombuffer(foreign, dtype=np.uint8).copy()).cuda()
        d_dec.zero_()
        rc, ol, used, msg = ctx.decode_device(fmt, d_f.data_ptr(), len(foreign), d_dec.data_ptr(), n)
        assert rc == 0 and ol == n and used == len(foreign), ("decode foreign", case, n, rc, msg)
        assert d_dec[:n].cpu().numpy().tobytes() == data, ("decode foreign: bytes", case, n)
