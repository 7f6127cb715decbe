"""GPU box: encode 256 MiB TEXT a few times and print the match-stage phase time (LFX_R7_CAP experiments: the output is not checked)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import ctypes as C
import torch
import libflate_amd
from libflate_amd import _ffi
import synth
n = 256 << 20
ctx = libflate_amd.Context(0)
ctx.enable_timing(True)
d_in = torch.from_numpy(synth.text(n)).cuda()
opts, sched = _ffi.make_opts(mtime=0), _ffi.make_schedule(8192)
bound = _ffi.lib().lfx_encode_bound(n, C.byref(opts), C.byref(sched)) & ~3
d_out = torch.empty(bound, dtype=torch.uint8, device="cuda")
best = None
for r in range(4):
    m = ctx.encode_device(_ffi.GZIP, d_in.data_ptr(), n, d_out.data_ptr(), bound, opts, sched)
    t = dict(ctx.last_timing()["phases"])
    best = t if best is None or t["lz77_match"] < best["lz77_match"] else best
print("lz77_match=%.3f ms lz77_parse=%.3f ms m=%d" % (best["lz77_match"], best["lz77_parse"], m))
