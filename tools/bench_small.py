"""GPU box: latency of one gzip encode and one decode at small sizes (data resident in HBM, one stream, S8K writes).
Prints one JSON line per size: median of `reps` calls, wall clock around the blocking C-ABI call (every host round trip
of the control plane is inside).  Output is checked (decode == input; CRC trailer) once per size."""
import ctypes as C
import json
import os
import statistics
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import libflate_amd  # noqa: E402
import synth  # noqa: E402
from libflate_amd import _ffi  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [64 << 10, 1 << 20, 16 << 20, 64 << 20]
    reps = 15
    ctx = libflate_amd.Context(0)
    for n in sizes:
        data = synth.text(n)
        d_in = torch.from_numpy(data).cuda()
        opts, sched = _ffi.make_opts(mtime=0), _ffi.make_schedule(8192)
        bound = _ffi.lib().lfx_encode_bound(n, C.byref(opts), C.byref(sched)) & ~3
        d_out = torch.empty(bound, dtype=torch.uint8, device="cuda")
        d_dec = torch.empty(n, dtype=torch.uint8, device="cuda")
        te, td, m = [], [], 0
        for r in range(reps + 3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m = ctx.encode_device(_ffi.GZIP, d_in.data_ptr(), n, d_out.data_ptr(), bound, opts, sched)
            t1 = time.perf_counter()
            rc, ol, used, msg = ctx.decode_device(_ffi.GZIP, d_out.data_ptr(), m, d_dec.data_ptr(), n)
            t2 = time.perf_counter()
            assert rc == 0 and ol == n, (rc, msg)
            if r >= 3:
                te.append(t1 - t0)
                td.append(t2 - t1)
        assert torch.equal(d_dec, d_in)
        comp = d_out[:m].cpu().numpy().tobytes()
        assert int.from_bytes(comp[-8:-4], "little") == zlib.crc32(data.tobytes())
        e, d = statistics.median(te), statistics.median(td)
        # where the time goes: one more round trip with the library's phase events on (not part of the medians)
        ctx.enable_timing(True)
        ctx.encode_device(_ffi.GZIP, d_in.data_ptr(), n, d_out.data_ptr(), bound, opts, sched)
        pe = {k: round(v, 4) for k, v in (ctx.last_timing() or {"phases": []})["phases"]}
        ctx.decode_device(_ffi.GZIP, d_out.data_ptr(), m, d_dec.data_ptr(), n)
        pd = {k: round(v, 4) for k, v in (ctx.last_timing() or {"phases": []})["phases"]}
        ctx.enable_timing(False)
        print(json.dumps({"workload": "gzip TEXT S8K, one stream, resident in HBM", "bytes": n, "compressed_bytes": m,
                          "encode_ms": round(e * 1e3, 4), "decode_ms": round(d * 1e3, 4),
                          "encode_GBps": round(n / e / 1e9, 3), "decode_GBps": round(n / d / 1e9, 3), "reps": reps,
                          "encode_phases_ms": pe, "decode_phases_ms": pd}), flush=True)


main()
