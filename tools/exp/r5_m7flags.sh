#!/bin/bash
# round 5: lfx_match7 with progress counters instead of the per-tile barrier — a small case first (a hang must not hold the box),
# then the corpus, the stress and the timings of both builds
cd $GRAFT_REPO_ROOT
timeout 60 python tools/exp/enc_timing.py 4194304 8192 1 2>&1 | grep -E "equal|rror|rep 0" | cut -c1-120 || { echo "SMALL CASE FAILED OR HUNG"; exit 1; }
timeout 120 python tools/exp/enc_timing.py 67108864 8192 1 2>&1 | grep -E "equal|rror" || { echo "64 MiB FAILED"; exit 1; }
timeout 600 python tools/exp/r3_diag.py 2>&1 | grep -vE "^==== env|amdgpu.ids" | head -40 | grep -v ": OK" ; echo "diag done"
timeout 300 python tools/exp/m5_stress.py 200 2>&1 | tail -3
timeout 120 python tools/exp/enc_timing.py 67108864 0 1 2>&1 | grep -E "equal|rror"
for i in 1 2; do timeout 300 python tools/exp/enc_timing.py 268435456 8192 4 2>&1 | grep -E "rep 3|rror" | cut -c1-200; done
LFX_SO=$PWD/libflate_amd/liblfx_bar.so timeout 300 python tools/exp/enc_timing.py 268435456 8192 4 2>&1 | grep -E "rep 3|rror" | cut -c1-200
LFX_DEBUG=1 timeout 300 python tools/exp/enc_timing.py 268435456 8192 1 2>&1 | grep -E "match7 wave" | head -16
