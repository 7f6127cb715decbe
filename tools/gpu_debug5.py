"""decode of foreign (python zlib / gzip) streams: where does the fast path stop, how fast is the result"""
import os, sys, time, zlib, gzip
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch
import __graft_entry__ as g
g.build()
import libflate_amd
from libflate_amd import _ffi
import synth
ctx = libflate_amd.Context(0)
ctx.enable_timing(True)
n = int(os.environ.get("N_MIB", "32")) << 20
data = synth.text(n)
raw = data.tobytes()
for name, comp in (("zlib-6", zlib.compress(raw, 6)), ("zlib-1", zlib.compress(raw, 1)), ("zlib-9", zlib.compress(raw, 9))):
    d_in = torch.frombuffer(bytearray(comp), dtype=torch.uint8).cuda()
    d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    for it in range(2):
        torch.cuda.synchronize()
        t = time.time()
        try:
            r = ctx.decode_device(_ffi.ZLIB, d_in.data_ptr(), len(comp), d_out.data_ptr(), n)
        except Exception as e:
            r = ("EXC", str(e))
        dt = time.time() - t
        os.environ.pop("LFX_DEBUG", None)
    ok = bool(torch.equal(d_out, torch.from_numpy(data).cuda()))
    print(name, len(comp), r[:3], "%.1f ms  %.3f GB/s" % (dt * 1e3, n / dt / 1e9), "equal", ok, [(k, round(v, 2)) for k, v in ctx.last_timing()["phases"]], flush=True)
    os.environ["LFX_DEBUG"] = os.environ.get("DBG", "")
    if not os.environ["LFX_DEBUG"]:
        os.environ.pop("LFX_DEBUG")
