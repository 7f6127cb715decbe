#!/bin/bash
# quick check after a kernel change: parity on the diagnostic corpus, phase timings
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 600 python tools/exp/r3_diag.py 2>&1 | grep -vE "^==== env|amdgpu.ids" | head -16 | grep -v ": OK" ; echo "diag done"
timeout 300 python tools/exp/enc_timing.py 67108864 8192 1 2>&1 | grep -E "equal|rror"
timeout 300 python tools/exp/enc_timing.py 268435456 8192 4 2>&1 | grep -E "rep 3" | cut -c1-330
