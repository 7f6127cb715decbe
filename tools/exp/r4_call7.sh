#!/bin/bash
cd $GRAFT_REPO_ROOT
for cap in 0 262 260 259 258 4 2; do LFX_MATCH_V3=1 LFX_M3_CAP=$cap timeout 100 python tools/exp/m3_cap.py 2>/dev/null | tail -1; done
