#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/kt_m6
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_m6 -- python $R/bench.py --child --steps 3 --warmup 1 > $O/kt_m6.log 2>&1
f=$(find $O/kt_m6 -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(r['Name'][:60].ljust(60), r['Calls'], round(float(r['AverageNs'])/1e6,4))
PY
grep -a "walk_finish" $f | cut -c1-40,200-330
rm -rf $O/kt_m6
