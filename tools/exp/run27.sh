#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 200 python tools/exp/enc_timing.py 268435456 8192 3 2>&1 | grep "rep 2" | tr '|' '\n' | tail -1
LFX_DEBUG=1 timeout 100 python tools/exp/enc_timing.py 268435456 8192 1 2>&1 | grep -E "finder:|not-ok" | head -3
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py tests/test_gpu_round2.py -x -q -m gpu 2>&1 | tail -3
