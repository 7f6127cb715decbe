#!/bin/bash
# GPU box: which of its two speeds does the storing scan take?  N processes per setting of an environment switch
cd $GRAFT_REPO_ROOT
for v in "" "$@"; do
  echo "== [$v]"
  for i in 1 2 3 4 5 6; do
    env $v python bench.py --steps 5 --warmup 2 --no-subs --no-traffic --no-cpu-baseline --no-s1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); p=d['phases_ms']; print(d['value'], d['decode_GBps'], 'scan', p['dec:blk_scan'], 'copy', p['dec:lz77_copy'], 'place', p['dec:blk_emit'])"
  done
done
