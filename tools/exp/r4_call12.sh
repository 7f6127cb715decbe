#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out
LFX_DEBUG=1 timeout 100 python tools/exp/m3_cap.py 2> $O/m5v4_dbg.err | tail -1
grep -a "match3 wave[0-9]*:" $O/m5v4_dbg.err | tail -16 | cut -c1-120 | awk 'NR<=3||NR%4==0'
grep -a "loop trips" $O/m5v4_dbg.err | tail -15 | awk 'NR%5==1' | cut -c1-200
grep -a "hops" $O/m5v4_dbg.err | tail -1
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_large.py tests/test_gpu_round4.py -m gpu -x -q > $O/par3.log 2>&1; grep -a "^E  \|^FAILED\|passed\|failed" $O/par3.log | cut -c1-250 | head -8
