"""GPU box: where does the round-3 encode front end go wrong?  Each case runs in its own process (a GPU fault must not
take the others down): LZ77 code words of ONE chunk through the plug-in against the oracle (first mismatch: index,
position, segment / lane), and whole streams against the oracle's bytes."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)


def data_of(kind, n):
    import numpy as np
    import synth
    if kind == "text":
        return synth.text(n).tobytes()
    if kind == "lowent":
        return synth.lowent(n).tobytes()
    if kind == "zeros":
        return bytes(n)
    return np.random.default_rng(5).integers(0, 256, n, dtype=np.uint8).tobytes()


def case_codes(kind, n):
    import ctypes as C
    import numpy as np
    import lfo_oracle as oracle
    import libflate_amd
    from libflate_amd import _ffi
    data = data_of(kind, n)
    ctx = libflate_amd.Context(0)
    st = C.c_int(0)
    h = _ffi.lib().lfx_lz77_new(ctx.handle, 32768, 258, C.byref(st))
    got = []

    def sink(_u, p, k):
        got.append(np.ctypeslib.as_array(p, shape=(k,)).copy())
    cb = _ffi.SINK_CB(sink)
    rc = _ffi.lib().lfx_lz77_encode(h, data, len(data), cb, None)
    rc2 = _ffi.lib().lfx_lz77_flush(h, cb, None)
    g = np.concatenate(got) if got else np.zeros(0, np.uint32)
    w = oracle.lz77_chunk(data)
    if len(g) == len(w) and (g == w).all():
        print("codes %s %d: OK (%d codes) rc=%d/%d" % (kind, n, len(w), rc, rc2))
        return
    m = min(len(g), len(w))
    k = int(np.argmax(g[:m] != w[:m])) if (g[:m] != w[:m]).any() else m
    steps = np.where((w[:k] & 0xFFFF) != 0, w[:k] >> 16, 1)
    pos = int(steps.sum())
    print("codes %s %d: MISMATCH at code %d of %d/%d, position %d = segment %d group %d offset %d; got %s want %s rc=%d/%d" % (
        kind, n, k, len(g), len(w), pos, pos // 3328, pos % 3328 // 52, pos % 52,
        [hex(int(x)) for x in g[k:k + 4]], [hex(int(x)) for x in w[k:k + 4]], rc, rc2))
    # how many mismatching codes, where the next agreement is
    bad = np.flatnonzero(g[:m] != w[:m])
    print("   mismatching codes: %d, first %s last %d" % (len(bad), bad[:8].tolist(), int(bad[-1]) if len(bad) else -1))


def case_stream(kind, n, ws):
    import zlib
    import lfo_oracle as oracle
    import libflate_amd
    from libflate_amd import _ffi
    data = data_of(kind, n)
    ctx = libflate_amd.Context(0)
    got = ctx.encode_host(_ffi.DEFLATE, data, _ffi.make_opts(), _ffi.make_schedule(ws))
    want = oracle.encode(oracle.DEFLATE, data, write_size=ws)
    if got == want:
        print("stream %s %d ws=%d: OK" % (kind, n, ws))
        return
    k = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), min(len(got), len(want)))
    try:
        back = zlib.decompress(got, -15)
        inf = "inflates to %d bytes, equal=%s" % (len(back), back == data)
    except Exception as e:
        inf = "does not inflate: %s" % e
    print("stream %s %d ws=%d: MISMATCH at byte %d of %d/%d; %s" % (kind, n, ws, k, len(got), len(want), inf))


CASES = [("codes", "text", 300000), ("codes", "text", 1 << 20), ("codes", "text", 4 << 20), ("codes", "text", 9 << 20),
         ("codes", "lowent", 1 << 20), ("codes", "zeros", 1 << 20), ("codes", "rand", 1 << 20),
         ("stream", "text", 1 << 20, 8192), ("stream", "text", 4 << 20, 8192), ("stream", "text", 24 << 20, 8192),
         ("stream", "lowent", 4 << 20, 8192), ("stream", "text", 100, 0), ("stream", "text", 53, 8192),
         ("stream", "text", 3329, 0), ("stream", "text", 66563, 8192)]

if __name__ == "__main__":
    if len(sys.argv) > 1:
        a = sys.argv[1:]
        if a[0] == "codes":
            case_codes(a[1], int(a[2]))
        else:
            case_stream(a[1], int(a[2]), int(a[3]))
        sys.exit(0)
    for env_extra in ({}, {"LFX_MATCH_V1": "1"}):
        print("==== env", env_extra, flush=True)
        for c in CASES:
            env = dict(os.environ, **env_extra)
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__)] + [str(x) for x in c], env=env, capture_output=True,
                                   text=True, timeout=240)
                out = (r.stdout.strip() or "(no output)") + ("" if r.returncode == 0 else "  [rc=%d] %s" % (r.returncode, r.stderr.strip()[-300:]))
            except subprocess.TimeoutExpired:
                out = "%s: TIMEOUT" % (c,)
            print(out, flush=True)
