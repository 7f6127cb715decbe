#!/bin/bash
mkdir -p gpurun_out
( LFX_NO_SERIAL=1 timeout 900 python -m pytest tests/test_gpu_round2.py -x -q -m gpu --durations=15 -k "total_bits or cfg5 or cfg2" 2>&1 | tail -40 ) > gpurun_out/r2_tests_b.log 2>&1
cat gpurun_out/r2_tests_b.log
