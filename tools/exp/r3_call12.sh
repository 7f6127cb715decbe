#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -s -k "virtual or windows" > $O/r3c12_t3.log 2>&1; tail -25 $O/r3c12_t3.log | cut -c1-300
timeout 1200 python -m pytest tests/test_gpu_round2.py -x -q -m gpu > $O/r3c12_t2.log 2>&1; tail -15 $O/r3c12_t2.log | cut -c1-300
