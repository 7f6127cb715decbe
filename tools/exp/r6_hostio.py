"""round 6: host-memory surface timings — pcie_inclusive (pageable / page-locked) and the io::copy protocol at several
stream-encoder batch sizes (LFX_ENC_BATCH_MB is read when a context is made).  Prints one JSON line per measurement."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import __graft_entry__ as g  # noqa: E402

g.build()
import bench  # noqa: E402
import libflate_amd  # noqa: E402
import synth  # noqa: E402
from libflate_amd import _ffi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256 << 20
data = synth.text(n, seed=synth.SEED_BASE + 2)
ctx = libflate_amd.Context(0)
print(json.dumps({"pcie": bench.sub_pcie(ctx, _ffi, data, 8192)}), flush=True)
ctx.close()
for mb in (0, 4, 16, 32, 64):
    if mb:
        os.environ["LFX_ENC_BATCH_MB"] = str(mb)
    c = libflate_amd.Context(0)
    r = bench.sub_stream_api(c, _ffi, data)
    print(json.dumps({"enc_batch_mb": mb or 8, "stream_api": {k: r[k] for k in ("value", "encode_GBps", "decode_GBps", "encode_ms", "decode_ms", "round_trip_ok", "encode_ms_in_batch_calls", "encode_batch_calls", "decode_ms_in_window_calls", "decode_window_calls", "host_memcpy_ms")}}), flush=True)
    c.close()
