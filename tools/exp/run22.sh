#!/bin/bash
# S1 (marker path) timing with the new and the old symbol kernel, then the suites that cover the marker path
cd $GRAFT_REPO_ROOT
timeout 200 python tools/exp/enc_timing.py 268435456 0 3 2>&1 | grep "rep 2" | tr '|' '\n' | tail -1
LFX_MAT_V1=1 timeout 200 python tools/exp/enc_timing.py 268435456 0 2 2>&1 | grep "rep 1" | tr '|' '\n' | tail -1
timeout 900 python -m pytest tests/test_gpu_large.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
