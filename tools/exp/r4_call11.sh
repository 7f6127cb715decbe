#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in 0x00 0x01 0x100 0x101 0x300 0x00; do echo -n "tune $t: "; LFX_M5_TUNE=$t timeout 100 python tools/exp/m3_cap.py 2>/dev/null | tail -1; done
