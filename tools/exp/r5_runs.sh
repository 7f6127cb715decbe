#!/bin/bash
# round 5: lfx_match7's run handling — parity (corpus incl. LOWENT / zeros, stress), cfg5 phases, the text timing
cd $GRAFT_REPO_ROOT
timeout 600 python tools/exp/r3_diag.py 2>&1 | grep -vE "^==== env|amdgpu.ids" | head -40 | grep -v ": OK" ; echo "diag done"
timeout 300 python tools/exp/m5_stress.py 300 2>&1 | tail -2
timeout 300 python tools/exp/enc_timing.py 67108864 8192 1 2>&1 | grep -E "equal|rror"
timeout 300 python tools/exp/enc_timing.py 268435456 8192 4 2>&1 | grep -E "rep 3|rror" | cut -c1-220
timeout 400 python tools/exp/cfg5_run.py 2>&1 | tail -3 | cut -c1-600
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
