#!/bin/bash
# match2 check: quick oracle check at 64 MiB, debug counters at 256 MiB, parity suite, in both modes
mkdir -p gpurun_out
export LFX_MATCH_V2=1
for mode in "LFX_NO_FUSED=1" "LFX_FUSED_MIN_CHUNKS=1"; do
  echo "=== $mode" 
  env $mode timeout 300 python tools/exp/enc_timing.py 67108864 8192 2 2>&1 | tail -3
  env $mode LFX_DEBUG=1 timeout 300 python tools/exp/enc_timing.py 268435456 8192 1 2>&1 | grep -E "match|rep"
  env $mode timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5
done > gpurun_out/r2_m2b.log 2>&1
cat gpurun_out/r2_m2b.log
