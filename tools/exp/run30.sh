#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 200 python tools/exp/enc_timing.py 268435456 8192 3 2>&1 | grep "rep 2" | tr '|' '\n' | sed -n 1,3p
timeout 200 python tools/bench_small.py 1048576 16777216 2>/dev/null | cut -c60-200
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
