import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import lfo_oracle as oracle
import libflate_amd
from libflate_amd import _ffi, lz77
rng = np.random.default_rng(42)
data = rng.integers(0, 256, 200000, dtype=np.uint8).tobytes()
ctx = libflate_amd.Context(0)
want = oracle.encode(oracle.DEFLATE, data, write_size=0)
for rep in range(6):
    got = ctx.encode_host(_ffi.DEFLATE, data, _ffi.make_opts(), _ffi.make_schedule(0))
    print(os.environ.get("LFX_MATCH_V3", "match5"), rep, len(got), len(want), got == want)
e = lz77.DefaultLz77Encoder()
sink = []
e.encode(data, sink); e.flush(sink)
w = oracle.lz77_chunk(data)
g = [lz77.Code.to_word(c) for c in sink]
pos = 0
for i, (a, b) in enumerate(zip(g, w)):
    if a != int(b):
        print("first diff: code", i, "pos", pos, lz77.Code.from_word(a), lz77.Code.from_word(int(b))); break
    pos += (a >> 16) if (a & 0xFFFF) else 1
else:
    print("plugin codes equal", len(g), len(w))
