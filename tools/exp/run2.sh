#!/bin/bash
# match2 bring-up: quick oracle check at 64 MiB in the three modes, then the parity suite in each mode
mkdir -p gpurun_out
for mode in "LFX_MATCH_V1=1" "LFX_NO_FUSED=1" "LFX_FUSED_MIN_CHUNKS=1"; do
  echo "=== $mode" 
  env $mode timeout 300 python tools/exp/enc_timing.py 67108864 8192 2 2>&1 | tail -4
done > gpurun_out/r2_m2_quick.log 2>&1
cat gpurun_out/r2_m2_quick.log
for mode in "LFX_NO_FUSED=1" "LFX_FUSED_MIN_CHUNKS=1"; do
  echo "=== $mode"
  env $mode timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15
done > gpurun_out/r2_m2_parity.log 2>&1
cat gpurun_out/r2_m2_parity.log
