#!/bin/bash
# round 5: the block finder's stages after a change — the candidate counts have to stay (TEXT 256 MiB, S8K: stage1=507036
# candidates=257), then the phase times, then parity
cd $GRAFT_REPO_ROOT
echo "$(LFX_DEBUG=1 timeout 300 python tools/exp/enc_timing.py 268435456 8192 1 2>&1 | grep -E 'finder:' | head -3 | tr '\n' ' ')"
echo "$(timeout 300 python tools/exp/enc_timing.py 268435456 8192 5 2>&1 | grep -E 'rep 4|rror' | sed 's/.*| dec //' | cut -c1-200)"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_large.py -x -q -m gpu 2>&1 | tail -2
