#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for mode in "LFX_MATCH_V1=1" "LFX_NO_FUSED=1" "LFX_FUSED_MIN_CHUNKS=1"; do
  tag=$(echo $mode | cut -d= -f1)
  rm -rf $R/gpurun_out/pmc_$tag
  env $mode rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_$tag -- python $R/tools/exp/enc_timing.py 268435456 8192 1 > $R/gpurun_out/pmc_$tag.log 2>&1
  echo "=== $mode"
  python $R/tools/pmc_summary.py $R/gpurun_out/pmc_$tag | grep -E "lz77|parse|Name"
done > $R/gpurun_out/r2_pmc.log 2>&1
cat $R/gpurun_out/r2_pmc.log
