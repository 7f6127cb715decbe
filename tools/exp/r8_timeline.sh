#!/bin/bash
# GPU box: kernel timeline of one step of the bare bench loop
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$PWD; out=$R/gpurun_out/${1:-r8l}; mkdir -p $out
cd /tmp; rm -rf /tmp/kt_tl
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt_tl -- python $R/bench.py --child --steps 5 --warmup 2 > /tmp/kt_tl.log 2>&1
python $R/tools/timeline.py /tmp/kt_tl > $out/timeline.csv 2>$out/timeline.err; cat $out/timeline.csv; cat $out/timeline.err | tail -3
