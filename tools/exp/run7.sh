#!/bin/bash
mkdir -p gpurun_out
( timeout 300 python tools/exp/enc_timing.py 268435456 8192 3 2>&1 | grep rep | cut -d'|' -f1,2
  timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
  LFX_FUSED_MIN_CHUNKS=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3 ) > gpurun_out/r2_m2d.log 2>&1
cat gpurun_out/r2_m2d.log
