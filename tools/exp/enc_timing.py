"""GPU box: phase timings of one gzip encode + decode (256 MiB TEXT by default), optional LFX_DEBUG counters."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256 << 20
    ws = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    import libflate_amd
    from libflate_amd import _ffi
    import synth
    ctx = libflate_amd.Context(0)
    ctx.enable_timing(True)
    data = synth.text(n)
    d_in = torch.from_numpy(data).cuda()
    opts, sched = _ffi.make_opts(mtime=0), _ffi.make_schedule(ws)
    bound = _ffi.lib().lfx_encode_bound(n, C.byref(opts), C.byref(sched)) & ~3
    d_out = torch.empty(bound, dtype=torch.uint8, device="cuda")
    d_dec = torch.empty(n, dtype=torch.uint8, device="cuda")
    for r in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = ctx.encode_device(_ffi.GZIP, d_in.data_ptr(), n, d_out.data_ptr(), bound, opts, sched)
        t1 = time.perf_counter()
        te = ctx.last_timing()
        rc, ol, used, msg = ctx.decode_device(_ffi.GZIP, d_out.data_ptr(), m, d_dec.data_ptr(), n)
        t2 = time.perf_counter()
        td = ctx.last_timing()
        assert rc == 0 and ol == n, (rc, msg)
        print("rep %d: n=%d m=%d enc %.3f ms dec %.3f ms | enc %s | dec %s" % (
            r, n, m, (t1 - t0) * 1e3, (t2 - t1) * 1e3,
            " ".join("%s=%.3f" % (k, v) for k, v in te["phases"]),
            " ".join("%s=%.3f" % (k, v) for k, v in td["phases"])), flush=True)
    assert torch.equal(d_dec, d_in)
    import zlib
    if n <= (64 << 20):
        import lfo_oracle as oracle
        want = oracle.encode(oracle.GZIP, data.tobytes(), write_size=ws)
        got = d_out[:m].cpu().numpy().tobytes()
        print("oracle equal:", got == want, flush=True)
    else:
        got = d_out[:m].cpu().numpy().tobytes()
        print("crc ok:", zlib.crc32(data.tobytes()) == int.from_bytes(got[-8:-4], "little"), flush=True)


main()
