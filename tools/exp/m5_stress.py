"""GPU box: stress of the candidate stage through the Lz77Encode plug-in: many chunks of mixed kinds in ONE context (so that
scratch left by one chunk is garbage for the next), codes against the oracle's, first difference with its position."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

import lfo_oracle as oracle  # noqa: E402
import libflate_amd  # noqa: E402
import synth  # noqa: E402
from libflate_amd import lz77  # noqa: E402

rng = np.random.default_rng(7)
text = synth.text(600000).tobytes()
low = synth.lowent(600000).tobytes()


def sample(n, kind):
    if kind == 0:
        o = int(rng.integers(0, len(text) - n + 1)); return text[o:o + n]
    if kind == 1:
        o = int(rng.integers(0, len(low) - n + 1)); return low[o:o + n]
    if kind == 2:
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == 3:
        return rng.integers(0, int(rng.integers(2, 6)), n, dtype=np.uint8).tobytes()
    period = int(rng.integers(1, 400))
    unit = rng.integers(0, 256, period, dtype=np.uint8).tobytes()
    return (unit * (n // period + 1))[:n]


bad = 0
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for t in range(trials):
    n = int(rng.choice([262144, 262145, 200000, 100000, 65536, 40000, 300001]))
    kind = int(rng.integers(0, 5))
    data = sample(n, kind)
    e = lz77.DefaultLz77Encoder()
    sink = []
    e.encode(data[:262143], sink)
    e.flush(sink)
    want = oracle.lz77_chunk(data[:262143])
    got = np.array([lz77.Code.to_word(c) for c in sink], dtype=np.uint32)
    if len(got) == len(want) and (got == want).all():
        continue
    bad += 1
    pos = 0
    for i in range(min(len(got), len(want))):
        g, w = int(got[i]), int(want[i])
        if g != w:
            print("trial %d kind %d n %d: code %d at pos %d: got %s want %s" % (t, kind, n, i, pos, lz77.Code.from_word(g), lz77.Code.from_word(w)))
            break
        pos += (g >> 16) if (g & 0xFFFF) else 1
print("trials %d, mismatching %d" % (trials, bad))
