"""round 6: where a stream-decoder window's time goes (LFX_DEBUG lines of dec_gpu) + the phases of the last window"""
import os
import sys
os.environ["LFX_DEBUG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np
import __graft_entry__ as g
g.build()
import libflate_amd, synth, stream_copy
from libflate_amd import _ffi
n = 128 << 20
data = synth.text(n, seed=synth.SEED_BASE + 2)
ctx = libflate_amd.Context(0)
ctx.enable_timing(True)
enc = np.zeros(n, dtype=np.uint8); dec = np.zeros(n, dtype=np.uint8)
rc, m, te = stream_copy.encode(ctx, _ffi.GZIP, _ffi.make_opts(mtime=0), data.ctypes.data, n, 8192, enc.ctypes.data, enc.size)
for rep in range(2):
    rc, ol, td = stream_copy.decode(ctx, _ffi.GZIP, enc.ctypes.data, m, 8192, dec.ctypes.data, n)
    print("decode", rc, ol, "%.2f ms" % (td * 1e3), stream_copy.last_split(), file=sys.stderr)
