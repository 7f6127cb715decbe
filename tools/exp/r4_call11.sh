#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in 0x00 0x10 0x20 0x30 0x02 0x12 0x22 0x112 0x212; do echo -n "tune $t: "; LFX_M5_TUNE=$t timeout 100 python tools/exp/m3_cap.py 2>/dev/null | tail -1; done
