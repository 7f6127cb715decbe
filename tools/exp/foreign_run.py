"""GPU box: decode of ONE large stream made by another encoder (python zlib, level 6: ~60 KB blocks that read the 32 KiB in
front of them — the marker path) with the library's phase events.  Usage: foreign_run.py [bytes] [reps]"""
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import libflate_amd  # noqa: E402
import synth  # noqa: E402
from libflate_amd import _ffi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128 << 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = libflate_amd.Context(0)
ctx.enable_timing(True)
data = synth.text(n, seed=synth.SEED_BASE + 7)
z = zlib.compress(data.tobytes(), 6)
d_z = torch.from_numpy(np.frombuffer(z, dtype=np.uint8).copy()).cuda()
d_dec = torch.empty(n, dtype=torch.uint8, device="cuda")
best, pd = None, {}
for _ in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc, ol, used, msg = ctx.decode_device(_ffi.ZLIB, d_z.data_ptr(), len(z), d_dec.data_ptr(), n)
    dt = time.perf_counter() - t0
    if best is None or dt < best:
        best, pd = dt, {k: round(v, 4) for k, v in ctx.last_timing()["phases"]}
ok = rc == 0 and ol == n and torch.equal(d_dec, torch.from_numpy(data).cuda())
print(json.dumps({"workload": "python zlib level 6 stream of TEXT(%d B): %d B" % (n, len(z)), "decode_ms": round(best * 1e3, 3),
                  "decode_GBps_of_output": round(n / best / 1e9, 2), "ok": ok, "phases_ms": pd}))
