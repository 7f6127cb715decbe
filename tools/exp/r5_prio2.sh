#!/bin/bash
# round 5: rotating wavefront priority in the symbol kernels (LFX_DEC_PRIO=0: liblfx_a.so) — decode phases, then parity
cd $GRAFT_REPO_ROOT
echo "prio off: $(LFX_SO=$PWD/libflate_amd/liblfx_a.so timeout 300 python tools/exp/enc_timing.py 268435456 8192 5 2>&1 | grep -E 'rep 4|rror' | sed 's/.*| dec //' | cut -c1-200)"
echo "prio on:  $(timeout 300 python tools/exp/enc_timing.py 268435456 8192 5 2>&1 | grep -E 'rep 4|rror' | sed 's/.*| dec //' | cut -c1-200)"
LFX_DEBUG=1 timeout 300 python tools/exp/enc_timing.py 268435456 8192 1 2>&1 | grep -E "K2 block" | head -3
for so in liblfx_a.so liblfx.so; do
  LFX_SO=$PWD/libflate_amd/$so timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-s1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('cfg3_batch_decode',{})
print('$so', 'value', d['value'], 'cfg3', {k: s.get(k) for k in ('value','decode_GBps','encode_GBps','decode_ms','encode_ms')}, 'cfg5', {k: d.get('cfg5_lowent_encode',{}).get(k) for k in ('encode_GBps','decode_ms')})"
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round4.py -x -q -m gpu 2>&1 | tail -2
