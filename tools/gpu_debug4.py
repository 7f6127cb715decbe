"""code-level comparison of the LZ77 plug-in path with the oracle on one chunk"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np
import torch
os.environ["LFX_DEBUG"] = "1"
import __graft_entry__ as g
g.build()
import libflate_amd
from libflate_amd import lz77
import lfo_oracle as O, synth
for n in (5000, 70000, 262144):
    data = bytes(synth.text(n))
    want = O.lz77_chunk(data)
    enc = lz77.DefaultLz77Encoder()
    sink = []
    enc.encode(data, sink)
    enc.flush(sink)
    got = np.array([(c[1] << 16) if c[0] == "Literal" else (c[1] << 16 | c[2]) for c in sink], dtype=np.uint32)
    k = min(len(got), len(want))
    neq = np.nonzero(got[:k] != want[:k])[0]
    print(n, "codes", len(got), len(want), "first mismatch", (int(neq[0]) if len(neq) else None))
    if len(neq):
        i = int(neq[0])
        pos = int(np.where((want[:i] & 0xFFFF) > 0, want[:i] >> 16, 1).sum())
        def fmt(a, lo, hi, p0):
            out = []
            p = p0
            for x in a[lo:hi]:
                x = int(x)
                out.append("%d:%s" % (p, ("L%02x" % (x >> 16)) if (x & 0xFFFF) == 0 else "M(%d,%d)" % (x >> 16, x & 0xFFFF)))
                p += (x >> 16) if (x & 0xFFFF) else 1
            return " ".join(out)
        lo = max(0, i - 6)
        p0 = int(np.where((want[:lo] & 0xFFFF) > 0, want[:lo] >> 16, 1).sum())
        print("  got ", fmt(got, lo, i + 8, p0))
        print("  want", fmt(want, lo, i + 8, p0))
        break
