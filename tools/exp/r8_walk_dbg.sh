#!/bin/bash
# GPU box: cycle stamps of parse_walk (LFX_DEBUG) on the text and on LOWENT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r8a
LFX_DEBUG=1 timeout 300 python tools/exp/enc_timing.py 268435456 8192 1 2>&1 | grep -E "walk wave|^rep" > gpurun_out/r8a/text.txt
LFX_DEBUG=1 timeout 300 python tools/exp/cfg5_run.py 268435456 1 2>&1 | grep -E "walk wave|workload" > gpurun_out/r8a/lowent.txt
cat gpurun_out/r8a/text.txt gpurun_out/r8a/lowent.txt
