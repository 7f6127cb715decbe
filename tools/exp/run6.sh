#!/bin/bash
mkdir -p gpurun_out
for mode in "LFX_MATCH_V1=1" "LFX_MATCH_V2=1 LFX_NO_FUSED=1" "LFX_MATCH_V2=1"; do
  echo "=== $mode" 
  env $mode timeout 300 python tools/exp/enc_timing.py 268435456 8192 3 2>&1 | grep rep | cut -d'|' -f1,2
done > gpurun_out/r2_m2c.log 2>&1
cat gpurun_out/r2_m2c.log
