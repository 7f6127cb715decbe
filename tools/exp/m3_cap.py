"""GPU box: timing-only experiment on lz77_match3_kernel (LFX_DEBUG + LFX_M3_CAP): what the kernel would cost if its
phase-B loop ended after N trips.  Encodes only; the output of a capped run is WRONG by construction."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import libflate_amd  # noqa: E402
import synth  # noqa: E402
from libflate_amd import _ffi  # noqa: E402

n = 256 << 20
ctx = libflate_amd.Context(0)
ctx.enable_timing(True)
d_in = torch.from_numpy(synth.text(n)).cuda()
opts, sched = _ffi.make_opts(mtime=0), _ffi.make_schedule(8192)
bound = _ffi.lib().lfx_encode_bound(n, C.byref(opts), C.byref(sched)) & ~3
d_out = torch.empty(bound, dtype=torch.uint8, device="cuda")
best = None
for r in range(3):
    m = ctx.encode_device(_ffi.GZIP, d_in.data_ptr(), n, d_out.data_ptr(), bound, opts, sched)
    ph = dict(ctx.last_timing()["phases"])
    best = ph["lz77_match"] if best is None else min(best, ph["lz77_match"])
print("LFX_M3_CAP=%s: lz77_match %.3f ms (compressed %d)" % (os.environ.get("LFX_M3_CAP", "-"), best, m))
