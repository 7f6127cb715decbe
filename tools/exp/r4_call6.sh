#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 100 python tools/exp/m3_cap.py 2>/dev/null | tail -1
LFX_MATCH_V3=1 timeout 100 python tools/exp/m3_cap.py 2>/dev/null | tail -1
LFX_DEBUG=1 timeout 100 python tools/exp/m3_cap.py 2> $O/m5_dbg.err | tail -1
grep -a "hops\|match3 wave[0-9]*:" $O/m5_dbg.err | tail -18 | cut -c1-200
grep -a "loop trips" $O/m5_dbg.err | tail -15 | awk 'NR%4==1' | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round2.py -m gpu -x -q 2>&1 | tail -6
