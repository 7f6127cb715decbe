#!/bin/bash
cd $GRAFT_REPO_ROOT
LFX_DEBUG=1 timeout 200 python tools/bench_small.py 8192 65536 1048576 2> gpurun_out/r4_dbg_small.err >/dev/null
grep -n "piece \|K2 block 0\|K3 unit 0\|K3: units\|pieces:" gpurun_out/r4_dbg_small.err | awk -F: '{print $2": "$3" "$4}' | sort | uniq -c | sort -rn | head -40
