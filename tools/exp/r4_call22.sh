#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for t in 0 1 2 3 4 7; do
rm -rf $O/kt_m6
LFX_M6_TUNE=$t timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_m6 -- python $R/tools/exp/m3_cap.py > $O/kt_m6.log 2>&1
f=$(find $O/kt_m6 -name "*kernel_stats.csv" | head -1); echo -n "tune $t: "; grep -a "lz77_match6\|walk_finish" $f | awk -F, '{printf "%s ", $(NF-4)}'; echo
done
rm -rf $O/kt_m6
