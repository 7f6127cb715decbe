"""First-contact diagnostics on the GPU box: runs the stages one by one and prints where the HIP path
first diverges from the oracle (more informative than a bare pytest failure)."""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402


def main():
    import __graft_entry__ as g
    g.build()
    import libflate_amd
    from libflate_amd import _ffi
    import lfo_oracle as oracle
    import synth
    ctx = libflate_amd.Context(0)
    ctx.enable_timing(True)

    def step(name, fn):
        t = time.time()
        try:
            r = fn()
            print("[ok ] %-40s %.2fs %s" % (name, time.time() - t, r if r is not None else ""), flush=True)
        except Exception:
            print("[ERR] %-40s" % name, flush=True)
            traceback.print_exc()

    text = synth.text(1 << 20).tobytes()

    def lz_small():
        z = libflate_amd.lz77.DefaultLz77Encoder.new()
        s = []
        z.encode(b"aaaaa", s); z.flush(s)
        return s
    step("lz77 aaaaa", lz_small)

    def lz_text():
        z = libflate_amd.lz77.DefaultLz77Encoder.new()
        s = []
        z.encode(text[:200000], s); z.flush(s)
        want = [libflate_amd.lz77.Code.from_word(w) for w in oracle.lz77_chunk(text[:200000])]
        if s != want:
            for i, (a, b) in enumerate(zip(s, want)):
                if a != b:
                    return "MISMATCH at code %d: got %s want %s (n got %d want %d)" % (i, a, b, len(s), len(want))
            return "MISMATCH length got %d want %d" % (len(s), len(want))
        return "codes %d" % len(s)
    step("lz77 text 200000", lz_text)

    def enc_case(fmt, data, ws, **kw):
        got = ctx.encode_host(fmt, data, _ffi.make_opts(**kw), _ffi.make_schedule(ws))
        want = oracle.encode(fmt, data, write_size=ws, **kw)
        if got != want:
            k = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), min(len(got), len(want)))
            return "MISMATCH first diff at byte %d of %d/%d" % (k, len(got), len(want))
        return "bytes %d" % len(got)
    step("deflate hello", lambda: enc_case(_ffi.DEFLATE, b"Hello World!", 0))
    step("deflate empty", lambda: enc_case(_ffi.DEFLATE, b"", 0))
    step("gzip text 1MiB S8K", lambda: enc_case(_ffi.GZIP, text, 8192))
    step("zlib text 1MiB S1", lambda: enc_case(_ffi.ZLIB, text, 0))
    step("zlib stored", lambda: enc_case(_ffi.ZLIB, text[:200000], 0, no_compression=1))
    step("zlib fixed", lambda: enc_case(_ffi.ZLIB, text[:200000], 8192, dynamic_huffman=0))
    print("timing:", ctx.last_timing(), flush=True)

    def dec_case(fmt, stream, want):
        rc, out, used, msg = ctx.decode_host(fmt, stream)
        if (rc, out, used) != (0, want, len(stream)):
            k = next((i for i in range(min(len(out), len(want))) if out[i] != want[i]), min(len(out), len(want)))
            return "MISMATCH rc=%d msg=%r out=%d want=%d firstdiff=%d used=%d/%d" % (rc, msg, len(out), len(want), k, used, len(stream))
        return "out %d" % len(out)
    step("decode fixed hello", lambda: dec_case(_ffi.DEFLATE, bytes([243, 72, 205, 201, 201, 87, 8, 207, 47, 202, 73, 81, 4, 0]), b"Hello World!"))
    step("decode oracle gzip 1MiB S8K", lambda: dec_case(_ffi.GZIP, oracle.encode(oracle.GZIP, text, 8192), text))
    import zlib
    step("decode python zlib 1MiB", lambda: dec_case(_ffi.ZLIB, zlib.compress(text, 6), text))
    big = synth.text(8 << 20).tobytes()
    step("decode oracle gzip 8MiB S8K (finder)", lambda: dec_case(_ffi.GZIP, oracle.encode(oracle.GZIP, big, 8192), big))
    print("timing:", ctx.last_timing(), flush=True)

    def big_enc():
        import torch
        n = 256 << 20
        data = synth.text(n)
        d_in = torch.from_numpy(data).cuda()
        bound = _ffi.lib().lfx_encode_bound(n, None, None)
        d_out = torch.empty(bound, dtype=torch.uint8, device="cuda")
        sched, opts = _ffi.make_schedule(8192), _ffi.make_opts()
        for it in range(3):
            torch.cuda.synchronize()
            t = time.time()
            m = ctx.encode_device(_ffi.GZIP, d_in.data_ptr(), n, d_out.data_ptr(), bound, opts, sched)
            dt = time.time() - t
            print("  encode 256MiB: %.2f ms  %.2f GB/s  C/N %.3f" % (dt * 1e3, n / dt / 1e9, m / n), ctx.last_timing(), flush=True)
        d_dec = torch.zeros(n, dtype=torch.uint8, device="cuda")
        for it in range(2):
            torch.cuda.synchronize()
            t = time.time()
            rc, ol, used, msg = ctx.decode_device(_ffi.GZIP, d_out.data_ptr(), m, d_dec.data_ptr(), n)
            dt = time.time() - t
            print("  decode 256MiB: rc=%d %.2f ms  %.2f GB/s" % (rc, dt * 1e3, n / dt / 1e9), ctx.last_timing(), msg, flush=True)
        return "roundtrip equal=%s" % bool(torch.equal(d_dec, d_in))
    step("256 MiB encode/decode timing", big_enc)


if __name__ == "__main__":
    main()
