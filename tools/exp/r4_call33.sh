#!/bin/bash
# round 4, last soak: randomized round trips at 1 .. 128 MiB (pieces over candidate ranges, repairs, windows, wide / narrow units) and 1000 more below 8 MiB
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
( echo "LFX_FUZZ=45 LFX_FUZZ_MINLOG2=20 LFX_FUZZ_MAXLOG2=27 LFX_FUZZ_SEED=3"; LFX_FUZZ=45 LFX_FUZZ_MINLOG2=20 LFX_FUZZ_MAXLOG2=27 LFX_FUZZ_SEED=3 timeout 150 python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | tail -3
  echo "LFX_FUZZ=1000 LFX_FUZZ_SEED=5"; LFX_FUZZ=1000 LFX_FUZZ_SEED=5 timeout 100 python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | tail -3 ) | tee $O/r4_fuzz2.txt
