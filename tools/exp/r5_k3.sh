#!/bin/bash
# round 5: K3's periodic source for tiles of long matches — parity (round-5 cases, fuzz, corpus), cfg5 phases, the text timing
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_round3.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/exp/enc_timing.py 268435456 8192 4 2>&1 | grep -E "rep 3|rror" | cut -c1-420
timeout 400 python tools/exp/cfg5_run.py 2>&1 | tail -1 | cut -c1-900
