#!/bin/bash
mkdir -p gpurun_out
( timeout 2300 python -m pytest tests/test_gpu_round2.py -x -q -m gpu 2>&1 | tail -40 ) > gpurun_out/r2_tests_a.log 2>&1
cat gpurun_out/r2_tests_a.log
