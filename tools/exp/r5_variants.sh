#!/bin/bash
# round 5: development builds of the library side by side (LFX_SO): decode phase times at 256 MiB
cd $GRAFT_REPO_ROOT
for t in ${VARIANTS:-a b c d}; do
  echo "variant $t: $(LFX_SO=$PWD/libflate_amd/liblfx_$t.so timeout 300 python tools/exp/enc_timing.py 268435456 8192 4 2>&1 | grep -E 'rep 3|rror' | sed 's/.*| dec //' | cut -c1-200)"
done
