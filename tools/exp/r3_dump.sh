#!/bin/bash
cd $GRAFT_REPO_ROOT
for sg in 1511 1510; do
  echo "== text 9M seg $sg"; LFX_DEBUG=1 LFX_DUMP_SEG=$sg timeout 200 python tools/exp/r3_diag.py codes text 9437184 2>&1 | grep -E "\[lfx\] seg|\[lfx\]  cd|codes text" | cut -c1-900
done
echo "== text 9M seg 1511 again (same process state?)"; LFX_DEBUG=1 LFX_DUMP_SEG=1511 timeout 200 python tools/exp/r3_diag.py codes text 9437184 2>&1 | grep -E "\[lfx\] seg|codes text" | cut -c1-400
for sg in 0 1 2; do
  echo "== zeros 1M seg $sg"; LFX_DEBUG=1 LFX_DUMP_SEG=$sg timeout 200 python tools/exp/r3_diag.py codes zeros 1048576 2>&1 | grep -E "\[lfx\] seg|\[lfx\]  cd|codes zeros|fault" | cut -c1-600
done
echo "== zeros 200000"; timeout 200 python tools/exp/r3_diag.py codes zeros 200000 2>&1 | tail -2
echo "== zeros 20000"; timeout 200 python tools/exp/r3_diag.py codes zeros 20000 2>&1 | tail -2
echo "== lowent 60000"; timeout 200 python tools/exp/r3_diag.py codes lowent 60000 2>&1 | tail -2
