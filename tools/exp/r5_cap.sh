#!/bin/bash
# round 5: what the resolver's long walks cost — every walk ended after LFX_R7_CAP hops (wrong answers, timing only)
cd $GRAFT_REPO_ROOT
for cap in 0 64 32 16 8 4 1; do
  echo "cap $cap: $(LFX_R7_CAP=$cap timeout 300 python tools/exp/r5_kt.py 2>&1 | tail -1)"
done
