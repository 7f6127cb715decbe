#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
rm -rf $R/gpurun_out/pmc_m2v
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_m2v -- python $R/tools/exp/enc_timing.py 268435456 8192 1 > $R/gpurun_out/pmc_m2v.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_m2v | grep -E "lz77|Name" > $R/gpurun_out/r2_pmc2.log
rm -rf $R/gpurun_out/pmc_m2w
rocprofv3 --pmc SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_m2w -- python $R/tools/exp/enc_timing.py 268435456 8192 1 > $R/gpurun_out/pmc_m2w.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_m2w | grep -E "lz77" >> $R/gpurun_out/r2_pmc2.log
cat $R/gpurun_out/r2_pmc2.log
