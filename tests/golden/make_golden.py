#!/usr/bin/env python3
"""Regenerates the binary fixtures under tests/golden/ from the reference tree.

Runs ONLY in the build container (needs /root/reference).  It copies DATA the reference's own
tests hold for the DEFLATE hot path — never source text:
  * data/issues_16/*                           (zlib.rs:798-837 reject vectors)
  * data/noncompressed_block_offset_sync/*     (non_blocking/gzip.rs:177-183 decode vector)
  * the byte values of ISSUE_52_INPUT          (src/deflate/test_data.rs:3-626, used by encode.rs:434-457)
"""
import os
import re
import shutil

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    for sub in ("issues_16", "noncompressed_block_offset_sync"):
        dst = os.path.join(HERE, sub)
        os.makedirs(dst, exist_ok=True)
        for f in os.listdir(os.path.join(REF, "data", sub)):
            shutil.copyfile(os.path.join(REF, "data", sub, f), os.path.join(dst, f))
    src = open(os.path.join(REF, "src/deflate/test_data.rs")).read()
    body = src[src.index("= [") + 3: src.rindex("]")]
    vals = [int(x) for x in re.findall(r"\d+", body)]
    assert len(vals) == 16052, len(vals)  # encode.rs:439-456 slices it at 16031 / 16032
    open(os.path.join(HERE, "issue_52_input.bin"), "wb").write(bytes(vals))
    print("issue_52_input.bin", len(vals))


if __name__ == "__main__":
    main()
