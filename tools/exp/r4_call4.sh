#!/bin/bash
# round 4, call 4: full GPU suite, small sizes, cfg5 profile (kernel stats + phases)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -q -m gpu -x > $O/r4_suite_a.log 2>&1; tail -4 $O/r4_suite_a.log
timeout 200 python tools/bench_small.py 65536 262144 1048576 4194304 16777216 67108864 > $O/r4_small_c.json 2>/dev/null; cut -c1-330 $O/r4_small_c.json
timeout 300 python tools/exp/cfg5_run.py > $O/r4_cfg5.json 2>$O/r4_cfg5.err; cat $O/r4_cfg5.json
cd /tmp; rm -rf $O/kt_cfg5
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg5 -- python $R/tools/exp/cfg5_run.py 1073741824 2 > $O/kt_cfg5.log 2>&1
f=$(find $O/kt_cfg5 -name "*kernel_stats.csv" | head -1); cp $f $O/r4_cfg5_kernel_stats.csv; rm -rf $O/kt_cfg5; head -14 $O/r4_cfg5_kernel_stats.csv | cut -c1-160
