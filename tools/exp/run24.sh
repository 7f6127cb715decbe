#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 200 python tools/exp/enc_timing.py 268435456 8192 3 2>&1 | grep "rep 2" | tr '|' '\n'
timeout 200 python tools/exp/enc_timing.py 268435456 0 3 2>&1 | grep "rep 2" | tr '|' '\n'
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
