#!/bin/bash
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_round2.py -x -q -m gpu --durations=12 2>&1 | tail -30 ) > gpurun_out/r2_tests_c.log 2>&1
cat gpurun_out/r2_tests_c.log
