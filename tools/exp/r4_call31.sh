#!/bin/bash
# pieces over candidate ranges with evenly split blocks and one symbol unit per piece: up to how many blocks does it pay?
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > $O/r4_suite_m.log 2>&1; tail -4 $O/r4_suite_m.log
for pm in 80 140; do
  echo "LFX_POCR_MAX=$pm"
  LFX_POCR_MAX=$pm timeout 200 python tools/bench_small.py 4194304 16777216 33554432 50331648 67108864 83886080 100663296 134217728 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(' ', d['bytes'], d['decode_ms'], d['decode_phases_ms'])"
done
