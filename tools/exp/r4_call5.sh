#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
for cap in 0 260 259 258 257 4 2; do
  LFX_DEBUG=1 LFX_M3_CAP=$cap timeout 120 python tools/exp/m3_cap.py 2> $O/m3_cap_$cap.err | tail -1
  grep "match3 wave[1-9]" $O/m3_cap_$cap.err | tail -32 | awk 'NR%16==3' | cut -c1-220
done
grep "match3 wave" $O/m3_cap_0.err | tail -31 > $O/r4_m3_trips.txt
