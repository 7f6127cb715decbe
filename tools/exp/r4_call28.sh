#!/bin/bash
# round 4 soak of the late kernels: randomized round trips (sizes 64 B .. 8 MiB, five data kinds, three containers, own and zlib streams)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
( LFX_FUZZ=400 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | tail -6
  LFX_FUZZ=200 LFX_FUZZ_SEED=7 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | tail -3 ) | tee $O/r4_fuzz.txt
