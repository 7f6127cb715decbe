#!/bin/bash
cd $GRAFT_REPO_ROOT
for so in liblfx_a.so liblfx_b.so liblfx.so liblfx_c.so liblfx_d.so; do
  echo "$so: $(LFX_SO=$PWD/libflate_amd/$so timeout 200 python tools/exp/enc_timing.py 268435456 8192 4 2>&1 | grep -E 'rep 3|rror' | sed 's/.*| dec //' | cut -c60-140)"
done
