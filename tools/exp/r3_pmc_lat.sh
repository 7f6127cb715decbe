#!/bin/bash
# latency counters per kernel: in-flight levels / instruction counts = average latency of vector-memory and LDS instructions
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out; mkdir -p $O
for pass in A B; do
  if [ $pass = A ]; then C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_LDS";
  else C="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU"; fi
  rm -rf $O/pmc_lat_$pass
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/pmc_lat_$pass -- python $R/bench.py --child --steps 1 --warmup 0 > $O/pmc_lat_$pass.log 2>&1
  echo "pass $pass rc=$?"; tail -2 $O/pmc_lat_$pass.log | cut -c1-200
  python $R/tools/pmc_summary.py $O/pmc_lat_$pass > $O/r3_pmc_lat_$pass.csv 2>/dev/null
  rm -rf $O/pmc_lat_$pass
done
cat $O/r3_pmc_lat_A.csv $O/r3_pmc_lat_B.csv | grep -v "at::native" | grep "blk_\|parse_walk\|match3\|find_" | sort | cut -c1-110
