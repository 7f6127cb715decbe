#!/bin/bash
# match kernel: time + instruction counts (the kernel is issue-bound: ACTIVE_INST_ANY == SIMD quad-cycles)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out; rm -rf $R/gpurun_out/pmc_i
( cd $R && timeout 300 python tools/exp/enc_timing.py 268435456 8192 3 2>&1 | grep "rep 2" | cut -d'|' -f2 | cut -c1-60
  timeout 300 python tools/exp/enc_timing.py 67108864 8192 1 2>&1 | grep -E "equal" )
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc_i -- python $R/tools/exp/enc_timing.py 268435456 8192 1 > $R/gpurun_out/pmc_i.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_i | grep -E "lz77_match" | awk -F, '{printf "%s=%.0fM ", $2, $4/1e6} END {print ""}'
cd $R && timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
