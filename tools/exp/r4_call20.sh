#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 100 python tools/exp/m3_cap.py 2>/dev/null | tail -1
LFX_MATCH_V5=1 timeout 100 python tools/exp/m3_cap.py 2>/dev/null | tail -1
timeout 600 python tools/exp/m5_stress.py 300 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r4_suite_g.log 2>&1; tail -3 $O/r4_suite_g.log
