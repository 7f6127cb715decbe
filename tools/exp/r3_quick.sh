#!/bin/bash
# quick check after a kernel change: parity on the diagnostic corpus, phase timings, optional cycle stamps
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 600 python tools/exp/r3_diag.py 2>&1 | grep -vE "^==== env|amdgpu.ids" | head -16 | grep -v ": OK" ; echo "diag done"
LFX_DEBUG=1 timeout 300 python tools/exp/enc_timing.py 268435456 8192 3 > $O/r3q_timing.log 2>&1; grep -E "rep 2|crc ok|Error|error" $O/r3q_timing.log | cut -c1-330; grep "match[34] wave" $O/r3q_timing.log | tail -16 | awk 'NR==1||NR==2||NR==5||NR==16'
timeout 300 python tools/exp/enc_timing.py 67108864 8192 1 2>&1 | grep -E "equal|rror"
timeout 300 python tools/exp/enc_timing.py 268435456 8192 3 2>&1 | grep -E "rep 2" | cut -c1-200
LFX_MATCH_NOMASK=1 timeout 300 python tools/exp/enc_timing.py 268435456 8192 3 2>&1 | grep -E "rep 2" | cut -c1-200
