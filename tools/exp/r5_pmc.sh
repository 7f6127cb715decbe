#!/bin/bash
# round 5: SQ counters of the kernels named in $K (default: the two Huffman passes of the decode) over one encode + decode of 256 MiB
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
K=${K:-"blk_scan_kernel|blk_emit_kernel|blk_materialize2|lz77_match7|parse_walk"}
for pass in 1 2 3; do
  if [ $pass = 1 ]; then C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS";
  elif [ $pass = 2 ]; then C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_BUSY_CYCLES";
  else C="SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE"; fi
  rm -rf /tmp/pmc_$pass
  timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$pass -- python $R/tools/exp/enc_timing.py 268435456 8192 1 > /tmp/pmc_$pass.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pmc_$pass 2>/dev/null | grep -E "$K" | cut -c1-140
done | tee $O/r5_pmc.txt
