"""GPU box: BASELINE cfg3 alone (bench.py's sub-record: 4096 x 64 KiB zlib streams, half reference-format, half python-zlib
level 6; one batch decode call, one batch encode call) — for a kernel trace of just this workload."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
import synth  # noqa: E402
import libflate_amd  # noqa: E402
from libflate_amd import _ffi  # noqa: E402

ctx = libflate_amd.Context(0)
ctx.enable_timing(True)
rec = bench.sub_cfg3(ctx, torch, synth, _ffi, C, torch.device("cuda:0"), reps=int(os.environ.get("REPS", "3")))
print(json.dumps(rec))
print(json.dumps(ctx.last_timing()))
