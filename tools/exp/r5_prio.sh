#!/bin/bash
# round 5: wavefront priorities by role in lfx_match7 (LFX_M7_PRIO tables), development builds side by side
cd $GRAFT_REPO_ROOT
for t in ${VARIANTS:-a b c d e}; do
  echo "variant $t: $(LFX_SO=$PWD/libflate_amd/liblfx_$t.so timeout 300 python tools/exp/enc_timing.py 268435456 8192 5 2>&1 | grep -E 'rep 4|rror' | sed 's/ | dec .*//' | cut -c30-200)"
done
