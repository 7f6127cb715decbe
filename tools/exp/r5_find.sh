#!/bin/bash
# round 5: the block finder's stage 2 in two passes (LFX_FIND2_CAP = steps of the first; 0 = one pass): the candidate counts
# have to agree, then the phase times, then parity
cd $GRAFT_REPO_ROOT
for cap in ${CAPS:-0 24 32 48 64 96}; do
  echo "cap $cap: $(LFX_FIND2_CAP=$cap LFX_DEBUG=1 timeout 300 python tools/exp/enc_timing.py 268435456 8192 1 2>&1 | grep -E 'finder:' | head -3 | tr '\n' ' ')"
  echo "cap $cap: $(LFX_FIND2_CAP=$cap timeout 300 python tools/exp/enc_timing.py 268435456 8192 5 2>&1 | grep -E 'rep 4|rror' | sed 's/.*| dec //' | cut -c1-200)"
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_large.py -x -q -m gpu 2>&1 | tail -2
