#!/usr/bin/env python3
"""Deterministic inputs for tools/ref_vectors (the REAL libflate, run by anyone with cargo) → tests/golden/ref/inputs/*.bin.

The reference's own tests pin encoder output only up to 48 input bytes; these inputs cover what the oracle
otherwise proves only against itself: whole 256 KiB LZ77 chunks and 1 MiB blocks (text, low-entropy runs, random bytes), the
chunk / block thresholds, the length-limiting branch of the Huffman builder (Fibonacci-skewed frequencies), and the
reference's own larger test inputs.  Small enough (about 9 MB) to commit the OUTPUT files next to kat.py once made."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import synth  # noqa: E402


def inputs():
    rng = np.random.default_rng(6)
    text = synth.text(3 << 20)
    yield "text_3mib", text.tobytes()
    yield "text_262143", text[:262143].tobytes()            # one byte below the chunk threshold (default.rs:60-68)
    yield "text_262144", text[:262144].tobytes()
    yield "text_1048577", text[:1048577].tobytes()          # one byte over the block size (encode.rs:277-286)
    yield "lowent_2mib", synth.lowent(2 << 20).tobytes()
    yield "random_300k", rng.integers(0, 256, 300000, dtype=np.uint8).tobytes()
    yield "zeros_300000", bytes(300000)
    yield "ramp_1mib", bytes(i & 255 for i in range(1 << 20))          # src/deflate/mod.rs:50-52
    yield "test_i", b"".join(b"test %d" % i for i in range(10000))     # src/non_blocking/deflate/decode.rs:274-277
    fib, parts = [1, 1], []
    while len(fib) < 30:
        fib.append(fib[-1] + fib[-2])
    for sym, f in enumerate(fib):
        parts.append(bytes([sym + 32]) * min(f, 400000))
    skew = np.frombuffer(b"".join(parts), dtype=np.uint8).copy()
    rng.shuffle(skew)
    yield "fib_skew", skew.tobytes()                        # unconstrained Huffman depth > 15: the package-merge limit bites


def main():
    out = os.path.join(os.path.dirname(HERE), "tests", "golden", "ref", "inputs")
    os.makedirs(out, exist_ok=True)
    for name, data in inputs():
        with open(os.path.join(out, name + ".bin"), "wb") as f:
            f.write(data)
        print(name, len(data))


if __name__ == "__main__":
    main()
