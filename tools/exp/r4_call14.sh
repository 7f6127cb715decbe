#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 100 python tools/exp/m3_cap.py 2>/dev/null | tail -1
timeout 300 python tools/exp/cfg5_run.py > $O/r4_cfg5_b.json 2>/dev/null; cut -c1-600 $O/r4_cfg5_b.json
timeout 600 python tools/exp/m5_stress.py 200 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r4_suite_c.log 2>&1; tail -3 $O/r4_suite_c.log
