"""GPU box: BASELINE cfg5 alone — zlib encode of 1 GiB LOWENT (8192-byte writes), `reps` encodes and one decode, with the
library's phase events printed.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel table
(tools/exp/r4_cfg5.sh keeps the summary under profiles/)."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import libflate_amd  # noqa: E402
import synth  # noqa: E402
from libflate_amd import _ffi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = libflate_amd.Context(0)
ctx.enable_timing(True)
data = synth.lowent(n, seed=synth.SEED_BASE + 5)
d_in = torch.from_numpy(data).cuda()
opts, sched = _ffi.make_opts(), _ffi.make_schedule(8192)
bound = _ffi.lib().lfx_encode_bound(n, C.byref(opts), C.byref(sched)) & ~3
d_out = torch.empty(bound, dtype=torch.uint8, device="cuda")
d_dec = torch.empty(n, dtype=torch.uint8, device="cuda")
best, m, ph = None, 0, {}
for _ in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m = ctx.encode_device(_ffi.ZLIB, d_in.data_ptr(), n, d_out.data_ptr(), bound, opts, sched)
    dt = time.perf_counter() - t0
    if best is None or dt < best:
        best, ph = dt, {k: round(v, 4) for k, v in ctx.last_timing()["phases"]}
torch.cuda.synchronize()
t0 = time.perf_counter()
rc, ol, used, msg = ctx.decode_device(_ffi.ZLIB, d_out.data_ptr(), m, d_dec.data_ptr(), n)
tdec = time.perf_counter() - t0
pd = {k: round(v, 4) for k, v in ctx.last_timing()["phases"]}
ok = rc == 0 and ol == n and torch.equal(d_dec, d_in)
print(json.dumps({"workload": "cfg5: zlib encode of LOWENT(%d B), 8192-byte writes" % n, "encode_ms": round(best * 1e3, 3),
                  "encode_GBps": round(n / best / 1e9, 3), "compressed_bytes": int(m), "decode_ms": round(tdec * 1e3, 3),
                  "round_trip_ok": ok, "encode_phases_ms": ph, "decode_phases_ms": pd}))
