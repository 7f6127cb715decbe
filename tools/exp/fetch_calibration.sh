#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/cal_$c
  timeout 120 rocprofv3 --pmc $c --output-format csv -d $O/cal_$c -- $R/tools/exp/fetch_cal > $O/cal_$c.log 2>&1
  python $R/tools/pmc_summary.py $O/cal_$c | grep -E "wide16|narrow4|strided4"
done
