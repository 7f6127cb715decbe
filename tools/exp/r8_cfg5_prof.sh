#!/bin/bash
# GPU box: per-kernel table of cfg5 (1 GiB LOWENT)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$PWD
out=$R/gpurun_out/${1:-r8j}; mkdir -p $out
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt_c5 -- python $R/tools/exp/cfg5_run.py > $out/cfg5_phases.json 2>$out/err.log
python $R/tools/prof_summary.py /tmp/kt_c5 "rocprofv3 --kernel-trace --stats -- python tools/exp/cfg5_run.py (zlib encode + decode of 1 GiB LOWENT, 8192-byte writes)" 2>/dev/null | grep -v "at::native\|elementwise" > $out/cfg5_kernel_stats.csv; head -24 $out/cfg5_kernel_stats.csv | cut -c1-120
