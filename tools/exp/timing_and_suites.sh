#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 200 python tools/exp/enc_timing.py 268435456 8192 3 2>&1 | grep "rep 2" | tr '|' '\n' | sed -n 1,2p
timeout 200 python tools/exp/enc_timing.py 67108864 8192 1 2>&1 | grep -E "equal"
timeout 200 python tools/exp/enc_timing.py 268435456 0 2 2>&1 | grep "rep 1" | tr '|' '\n' | sed -n 1,2p
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_large.py -x -q -m gpu 2>&1 | tail -3
