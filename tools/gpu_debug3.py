"""encode-path diagnostics on small inputs"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
os.environ["LFX_DEBUG"] = "1"
import torch
import __graft_entry__ as g
g.build()
import libflate_amd
from libflate_amd import _ffi
import synth
ctx = libflate_amd.Context(0)
ctx.enable_timing(True)
n = 64 << 20
data = synth.text(n)
d_in = torch.from_numpy(data).cuda()
bound = _ffi.lib().lfx_encode_bound(n, None, None) & ~3
d_out = torch.empty(bound, dtype=torch.uint8, device="cuda")
for it in range(2):
    m = ctx.encode_device(_ffi.GZIP, d_in.data_ptr(), n, d_out.data_ptr(), bound, _ffi.make_opts(), _ffi.make_schedule(8192))
    print("encode", m, ctx.last_timing(), flush=True)
