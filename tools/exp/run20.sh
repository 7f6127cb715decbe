#!/bin/bash
# phase timings (256 MiB S8K) + the bit-exactness suites
cd $GRAFT_REPO_ROOT
timeout 200 python tools/exp/enc_timing.py 268435456 8192 3 2>&1 | grep "rep 2" | tr '|' '\n'
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q -m gpu 2>&1 | tail -3
