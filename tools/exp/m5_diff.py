"""GPU box: the Lz77Encode plug-in's codes (match + parse on the GPU) against the oracle's, first differences."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

import lfo_oracle as oracle  # noqa: E402
import libflate_amd  # noqa: E402
from libflate_amd import lz77  # noqa: E402

rng = np.random.default_rng(42)
for _ in range(3):
    rng.integers(0, 256, 10, dtype=np.uint8)
data = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 42).integers(0, 256, 200000, dtype=np.uint8).tobytes()
for rep in range(4):
    e = lz77.DefaultLz77Encoder()
    sink = []
    e.encode(data[:150000], sink)         # < window * 8: buffered
    e.flush(sink)
    want = oracle.lz77_chunk(data[:150000])
    got = [lz77.Code.to_word(c) for c in sink]
    nd, pos = 0, 0
    for i, (g, w) in enumerate(zip(got, want)):
        if g != int(w):
            if nd < 5:
                print("rep %d code %d at pos %d: got %s want %s" % (rep, i, pos, lz77.Code.from_word(g), lz77.Code.from_word(int(w))))
            nd += 1
            break
        pos += (g >> 16) if (g & 0xFFFF) else 1
    print("rep %d: codes %d vs %d, first difference %s" % (rep, len(got), len(want), "none" if not nd else "above"))
