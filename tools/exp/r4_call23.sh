#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r4_suite_h.log 2>&1; tail -3 $O/r4_suite_h.log
bash tools/exp/r4_soak.sh > $O/r4_soak.txt 2>&1; cat $O/r4_soak.txt
