"""Oracle == the REAL libflate on vectors larger than the reference's own (nothing above 48 input bytes is
pinned there).  The vectors are made by tools/ref_vectors (Rust: needs cargo, which this image lacks) from the inputs of
tools/make_ref_inputs.py; until someone has run it, tests/golden/ref/ holds no outputs and this test SKIPS — it exists so
that the day the files appear, the oracle's one open pin closes without writing a line (README.md, "Reference vectors")."""
import glob
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "tests", "golden", "ref")


def _cases():
    return sorted(glob.glob(os.path.join(REF, "*.*.s*.bin")))


def test_oracle_equals_libflate_on_large_vectors(oracle):
    cases = _cases()
    if not cases:
        pytest.skip("no reference-made vectors under tests/golden/ref (run tools/ref_vectors with cargo)")
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ref_inputs
    data = dict(make_ref_inputs.inputs())
    fmts = {"deflate": oracle.DEFLATE, "zlib": oracle.ZLIB, "gzip": oracle.GZIP}
    for path in cases:
        name, fmt, sched = os.path.basename(path)[:-4].rsplit(".", 2)
        want = open(path, "rb").read()
        kw = {"mtime": 0} if fmt == "gzip" else {}
        got = oracle.encode(fmts[fmt], data[name], write_size=0 if sched == "s1" else 8192, **kw)
        assert got == want, (name, fmt, sched, len(got), len(want))
        rc, out, used, _ = oracle.decode(fmts[fmt], want)
        assert rc == 0 and out == data[name] and used == len(want), (name, fmt, sched)
