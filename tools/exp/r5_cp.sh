#!/bin/bash
# round 5: the scan's checkpoint (LFX_SCAN_CP_BITS: a = 384, default 768, b = 1536, c = 2560): decode phases at 256 MiB
cd $GRAFT_REPO_ROOT
for so in liblfx_a.so liblfx.so liblfx_b.so liblfx_c.so; do
  echo "$so: $(LFX_SO=$PWD/libflate_amd/$so timeout 300 python tools/exp/enc_timing.py 268435456 8192 5 2>&1 | grep -E 'rep 4|rror' | sed 's/.*| dec //' | cut -c1-200)"
done
