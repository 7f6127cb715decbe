#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 100 python tools/exp/m3_cap.py 2>/dev/null | tail -1
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -x -q 2>&1 | tail -2; done
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_large.py -m gpu -x -q 2>&1 | tail -2
