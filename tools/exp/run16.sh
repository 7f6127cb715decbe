#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out; rm -rf $R/gpurun_out/kt_m2
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_m2 -- python $R/tools/exp/enc_timing.py 268435456 8192 3 > $R/gpurun_out/kt_m2.log 2>&1
f=$(find $R/gpurun_out/kt_m2 -name "*kernel_stats.csv" | head -1)
head -12 $f | cut -c1-150
