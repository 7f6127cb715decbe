#!/bin/bash
# kernel stats of the default bench workload (cfg2 S8K): the bare timed loop of bench.py (--child), for profiles/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out; rm -rf $R/gpurun_out/kt_r2
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_r2 -- python $R/bench.py --child --steps 5 --warmup 2 > $R/gpurun_out/kt_r2.log 2>&1
f=$(find $R/gpurun_out/kt_r2 -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/kt_r2_kernel_stats.csv
head -40 $f | cut -d, -f1-7
