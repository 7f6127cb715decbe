#!/bin/bash
# decode: phase timings + LFX_DEBUG counters of the 256 MiB S8K stream, kernel stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 300 python tools/exp/enc_timing.py 268435456 8192 3 2>&1 | grep "rep 2" | tr '|' '\n'
LFX_DEBUG=1 timeout 300 python tools/exp/enc_timing.py 268435456 8192 1 2>&1 | grep -E "^\[lfx\]" | grep -v "cand \|match" | head -60
