#!/bin/bash
# pieces over candidate ranges: where does the limit belong now that the window resolution and the symbol kernel are cheaper?
cd $GRAFT_REPO_ROOT
for pm in 32 80 160 300; do
  echo "LFX_POCR_MAX=$pm"
  LFX_POCR_MAX=$pm timeout 200 python tools/bench_small.py 33554432 67108864 134217728 268435456 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(' ', d['bytes'], d['decode_ms'], d['decode_phases_ms'])"
done
