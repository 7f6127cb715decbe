#!/bin/bash
# round 5: cfg3 (batch decode / encode of 4096 x 64 KiB) under a kernel trace: which kernels its 3.6 ms are
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf /tmp/kt_c3
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt_c3 -- python $R/tools/exp/cfg3_run.py 2>/dev/null | tail -2 | cut -c1-1200
python $R/tools/prof_summary.py /tmp/kt_c3 "rocprofv3 --kernel-trace --stats -- python tools/exp/cfg3_run.py (4096 x 64 KiB zlib: one batch encode of 2048, 4 batch decodes of 4096)" 2>/dev/null | grep -v "at::native\|elementwise" > $O/r05_cfg3_kernel_stats.csv; head -24 $O/r05_cfg3_kernel_stats.csv | cut -c1-130
