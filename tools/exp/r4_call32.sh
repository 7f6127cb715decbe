#!/bin/bash
# pieces over candidate ranges: a range cut by a false candidate is scanned again (48 MiB of the corpus holds one)
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > $O/r4_suite_n.log 2>&1; tail -6 $O/r4_suite_n.log
timeout 200 python tools/bench_small.py 33554432 50331648 67108864 83886080 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(' ', d['bytes'], d['decode_ms'], d['decode_phases_ms'])"
