"""Static check of the compiled kernels for the pattern that cost this code base the most time, three rounds in a row: a
small loop with ONE global load and an `s_waitcnt vmcnt(0)` — every trip is a dependent HBM round trip (0.7–2 us), whatever
the loop computes.  (Round 3: `tile_bits`, `pack`, `parse_emit`, the checksum sweep; round 4: every step of the window
resolution — thirty-two round trips for a step that computes for one, 0.59 ms at 256 MiB.)

    python tools/isa_scan.py libflate_amd/csrc/lfx_inflate_fast.hip [kernel-name-substring ...]

compiles the file for gfx950 (device only, -S) and lists, per kernel, the inner loops of at most `max_len` instructions that
hold at most `max_loads` loads and a full wait.  Loops that run a handful of trips per launch (setup, stored blocks) are
expected in the list; a loop over the data is a finding."""
import os
import re
import subprocess
import sys
import tempfile

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def compile_to_asm(src):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-S", src, "-o", out],
                          stderr=subprocess.DEVNULL)
    with open(out) as f:
        text = f.read()
    os.unlink(out)
    return text


def serialized_load_loops(asm, max_loads=2, max_len=80):
    """→ [(kernel, label, instructions, loads, stores)] for the innermost loops (LLVM's own loop comments say which basic
    blocks belong to which loop header — a rotated loop's back edge does not point at its header)."""
    kern, cur = None, None
    inner, stat, order = set(), {}, []
    pending = None                      # label waiting for its (possibly next-line) loop comment
    for l in asm.split("\n"):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern, cur, pending = m.group(1), None, None
            continue
        lab = re.match(r"^\.(LBB\d+_\d+):", l)
        blk = re.match(r"^; %bb\.\d+:", l)
        cont = re.match(r"^\s+;", l) and pending is not None           # comment-only continuation of a label's note
        if lab or blk or cont:
            if lab:
                pending, cur = lab.group(1), None
            elif blk:
                pending, cur = "", None
            h = re.search(r"in Loop: Header=(BB\d+_\d+)", l)
            if h:
                cur = (kern, h.group(1))
            if "Loop Header" in l and pending:                         # this label IS a header
                cur = (kern, pending[1:])
                if "Inner Loop Header" in l:
                    inner.add(cur)
            if cur is not None and cur not in stat:
                stat[cur] = [0, 0, 0, 0]
                order.append(cur)
            continue
        if re.match(r"^\s*;", l) or not l.startswith("\t") or cur is None:
            continue
        pending = None
        st = stat[cur]
        st[0] += 1
        if re.search(r"\b(global|buffer|flat)_load", l):
            st[1] += 1
        if "global_store" in l:
            st[2] += 1
        if "s_waitcnt" in l and "vmcnt(0)" in l:
            st[3] += 1
    return [(k, "." + "L" + h, *stat[(k, h)][:3]) for (k, h) in order
            if (k, h) in inner and stat[(k, h)][1] and stat[(k, h)][3] and stat[(k, h)][1] <= max_loads and stat[(k, h)][0] < max_len]


if __name__ == "__main__":
    src, names = sys.argv[1], sys.argv[2:]
    for k, lab, n, loads, stores in serialized_load_loops(compile_to_asm(src)):
        if not names or any(s in k for s in names):
            print("%-70s %-10s %3d instructions, %d load(s), %d store(s), vmcnt(0)" % (k[:70], lab, n, loads, stores))
