#!/bin/bash
mkdir -p gpurun_out
for mode in "LFX_NO_FUSED=1" "LFX_FUSED_MIN_CHUNKS=1"; do
  echo "=== $mode"
  env $mode LFX_DEBUG=1 timeout 300 python tools/exp/enc_timing.py 268435456 8192 2 2>&1 | grep -E "match|rep"
done > gpurun_out/r2_m2_debug.log 2>&1
cat gpurun_out/r2_m2_debug.log
