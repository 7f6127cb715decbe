#!/bin/bash
mkdir -p gpurun_out
( timeout 300 python tools/exp/enc_timing.py 268435456 8192 3 2>&1 | grep rep | cut -d'|' -f1,2 | tail -1
  LFX_DEBUG=1 timeout 300 python tools/exp/enc_timing.py 268435456 8192 1 2>&1 | grep -E "match" | head -17 ) > gpurun_out/r2_m2f.log 2>&1
cat gpurun_out/r2_m2f.log
timeout 300 python tools/exp/enc_timing.py 67108864 8192 2 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
