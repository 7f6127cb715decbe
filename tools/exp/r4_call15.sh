#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 300 python tools/exp/cfg5_run.py > $O/r4_cfg5_c.json 2>/dev/null; cut -c1-420 $O/r4_cfg5_c.json
timeout 100 python tools/exp/m3_cap.py 2>/dev/null | tail -1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r4_suite_d.log 2>&1; tail -3 $O/r4_suite_d.log
