#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in ${VARIANTS:-a b}; do
LFX_SO=$PWD/libflate_amd/liblfx_$t.so LFX_DEBUG=1 timeout 300 python tools/exp/enc_timing.py 268435456 8192 1 2>&1 | grep -E "^\[lfx\]" | grep -v "match\|walk wave\|huffman" > gpurun_out/r5_dbg_$t.txt
echo "variant $t: cands $(grep -c ' cand ' gpurun_out/r5_dbg_$t.txt)"
grep " cand " gpurun_out/r5_dbg_$t.txt | awk '{for(i=1;i<=NF;i++){split($i,a,"="); if(a[1]=="status")s=a[2]; if(a[1]=="cyc_total")t=a[2]; if(a[1]=="cyc_hdr")h=a[2]; if(a[1]=="rounds")r=a[2]} n[s]++; tot[s]+=t; hd[s]+=h; if(t>mx[s])mx[s]=t; rr[s]+=r} END{for(s in n) print " status",s,"n",n[s],"avg_total",tot[s]/n[s],"avg_hdr",hd[s]/n[s],"max_total",mx[s],"avg_rounds",rr[s]/n[s]}'
grep "K2 block" gpurun_out/r5_dbg_$t.txt | head -3
done
