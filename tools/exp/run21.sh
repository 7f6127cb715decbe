#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 200 python tools/bench_small.py 2>&1 | tee gpurun_out/r2_small.json | tail -6
LFX_DEBUG=1 timeout 100 python tools/exp/enc_timing.py 1048576 8192 2 2>&1 | grep -E "rep 1|finder|chain" | head
