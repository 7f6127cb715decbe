#!/bin/bash
# first GPU call of round 2: instruction semantics + baseline counters
set -x
mkdir -p gpurun_out
./tools/exp/mskor_test > gpurun_out/r2_mskor.log 2>&1
python tools/exp/enc_timing.py 268435456 8192 3 > gpurun_out/r2_base_timing.log 2>&1
LFX_DEBUG=1 python tools/exp/enc_timing.py 67108864 8192 1 2>&1 | grep -E "match wave|rep 0|equal" > gpurun_out/r2_base_debug.log
tail -3 gpurun_out/r2_mskor.log gpurun_out/r2_base_timing.log
