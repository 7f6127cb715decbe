#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r4_suite_f.log 2>&1; tail -5 $O/r4_suite_f.log
timeout 200 python tools/bench_small.py 4194304 16777216 67108864 > $O/r4_small_e.json 2>/dev/null; cut -c1-190 $O/r4_small_e.json; python3 -c "
import json
for l in open('$O/r4_small_e.json'): d=json.loads(l); print(d['bytes'], d['decode_ms'], d['decode_phases_ms'])"
