#!/bin/bash
mkdir -p gpurun_out
cat > /tmp/abl.py <<'PY'
import os, sys, time, ctypes as C
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]+"/tools")
import torch, libflate_amd, synth
from libflate_amd import _ffi
ctx = libflate_amd.Context(0); ctx.enable_timing(True)
n = 256 << 20
d_in = torch.from_numpy(synth.text(n)).cuda()
opts, sched = _ffi.make_opts(mtime=0), _ffi.make_schedule(8192)
bound = _ffi.lib().lfx_encode_bound(n, C.byref(opts), C.byref(sched)) & ~3
d_out = torch.empty(bound, dtype=torch.uint8, device="cuda")
for r in range(3):
    try:
        ctx.encode_device(_ffi.GZIP, d_in.data_ptr(), n, d_out.data_ptr(), bound, opts, sched)
    except Exception as e:
        pass
    t = dict(ctx.last_timing()["phases"])
print(os.environ.get("LFX_ABLATE", "0"), "match=%.3f" % t["lz77_match"])
PY
for a in 0 1 2 4 8 16 32 3 7 39 47 63; do LFX_ABLATE=$a timeout 120 python /tmp/abl.py 2>&1 | tail -1; done > gpurun_out/r2_ablate.log 2>&1
cat gpurun_out/r2_ablate.log
