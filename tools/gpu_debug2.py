"""decode-path diagnostics on small inputs (cheap in GPU minutes)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
os.environ["LFX_DEBUG"] = "1"
os.environ["LFX_NO_SERIAL"] = "1"
import torch
import __graft_entry__ as g
g.build()
import libflate_amd
from libflate_amd import _ffi
import synth
ctx = libflate_amd.Context(0)
ctx.enable_timing(True)
for mib in (8, 256):
    n = mib << 20
    data = synth.text(n)
    d_in = torch.from_numpy(data).cuda()
    bound = _ffi.lib().lfx_encode_bound(n, None, None) & ~3
    d_out = torch.empty(bound, dtype=torch.uint8, device="cuda")
    m = ctx.encode_device(_ffi.GZIP, d_in.data_ptr(), n, d_out.data_ptr(), bound, _ffi.make_opts(), _ffi.make_schedule(8192))
    d_dec = torch.zeros(n, dtype=torch.uint8, device="cuda")
    for it in range(2):
        try:
            t = time.time()
            r = ctx.decode_device(_ffi.GZIP, d_out.data_ptr(), m, d_dec.data_ptr(), n)
            dt = time.time() - t
            print(mib, "MiB decode", r, "%.2f ms" % (dt * 1e3), "equal", bool(torch.equal(d_dec, d_in)), ctx.last_timing(), flush=True)
        except Exception as e:
            print(mib, "MiB decode EXC", e, ctx.last_timing(), flush=True)
        os.environ.pop("LFX_DEBUG", None)
    os.environ["LFX_DEBUG"] = "1"
