#!/bin/bash
cd $GRAFT_REPO_ROOT
for fs in 17 18 19; do echo "free_shift $fs"; LFX_FREE_SHIFT=$fs timeout 200 python tools/exp/enc_timing.py 268435456 0 2 2>&1 | grep "rep 1" | tr '|' '\n' | tail -1; done
echo default; timeout 200 python tools/exp/enc_timing.py 268435456 0 2 2>&1 | grep "rep 1" | tr '|' '\n' | tail -1
timeout 600 python -m pytest tests/test_gpu_large.py -x -q -m gpu 2>&1 | tail -2
