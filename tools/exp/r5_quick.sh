#!/bin/bash
# round 5: the check after a change of the candidate stage (lfx_match7.hip) — parity of the diagnostic corpus against the
# oracle, random chunk mixes, 64 MiB byte equality, phase timings at 256 MiB (both generations), cycle stamps of workgroup 0
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 600 python tools/exp/r3_diag.py 2>&1 | grep -vE "^==== env|amdgpu.ids" | head -40 | grep -v ": OK" ; echo "diag done"
timeout 300 python tools/exp/m5_stress.py ${STRESS:-120} 2>&1 | tail -5
timeout 300 python tools/exp/enc_timing.py 67108864 8192 1 2>&1 | grep -E "equal|rror"
timeout 300 python tools/exp/enc_timing.py 67108864 0 1 2>&1 | grep -E "equal|rror"
timeout 300 python tools/exp/enc_timing.py 268435456 8192 4 2>&1 | grep -E "rep 3|rror" | cut -c1-400
LFX_MATCH_V5=1 timeout 300 python tools/exp/enc_timing.py 268435456 8192 3 2>&1 | grep -E "rep 2|rror" | cut -c1-200
LFX_DEBUG=1 timeout 300 python tools/exp/enc_timing.py 268435456 8192 1 2>&1 | grep -E "match7 wave" | head -16
# per-kernel durations of the same loop
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/r5prof && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/r5prof -- python $GRAFT_REPO_ROOT/tools/exp/enc_timing.py 268435456 8192 5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py /tmp/r5prof 2>&1 | grep -v "at::native\|elementwise" | cut -c1-120 | head -30 | tee gpurun_out/r5_kernel_stats.csv
