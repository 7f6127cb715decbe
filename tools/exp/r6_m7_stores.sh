#!/bin/bash
# round 6: is lz77_match7_kernel bound by its stores?  Variants WITHOUT the link-record stores (1), the cd stores (2), both (3):
# wrong answers, timing only (encode alone, nothing is verified).  Built here, in the GPU call (hipcc is on the box).
cd "$GRAFT_REPO_ROOT" || exit 1
run() { LFX_R7_CAP=1 python - <<'PY'
import os, sys, ctypes as C
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"] + "/tools")
import torch, synth, libflate_amd
from libflate_amd import _ffi
n = 256 << 20
ctx = libflate_amd.Context(0); ctx.enable_timing(True)
d_in = torch.from_numpy(synth.text(n, seed=synth.SEED_BASE + 2)).cuda()
opts, sched = _ffi.make_opts(mtime=0), _ffi.make_schedule(8192)
bound = _ffi.lib().lfx_encode_bound(n, C.byref(opts), C.byref(sched)) & ~3
d_out = torch.empty(bound, dtype=torch.uint8, device="cuda")
best = None
for r in range(6):
    try:
        ctx.encode_device(_ffi.GZIP, d_in.data_ptr(), n, d_out.data_ptr(), bound, opts, sched)
    except Exception as e:
        print("encode raised", e); break
    t = dict(ctx.last_timing()["phases"])
    if best is None or t["lz77_match"] < best["lz77_match"]: best = t
print(os.environ.get("LFX_SO", "product"), {k: round(v, 4) for k, v in best.items() if k in ("lz77_match", "lz77_parse")})
PY
}
run
for v in 1 2 3; do LFX_SO=$GRAFT_REPO_ROOT/tools/exp/liblfx_m7exp$v.so run; done
