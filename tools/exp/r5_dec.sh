#!/bin/bash
# round 5: the check after a change of the decode kernels — full GPU suite, phase timings at 256 MiB (S8K and S1), cfg3 / cfg5
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 300 python tools/exp/enc_timing.py 268435456 8192 4 2>&1 | grep -E "rep 3|rror" | cut -c1-420
timeout 300 python tools/exp/enc_timing.py 268435456 0 3 2>&1 | grep -E "rep 2|rror" | cut -c1-520
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-s1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('bench', d['value'], 'GB/s enc', d['encode_GBps'], 'dec', d['decode_GBps'])
for k,v in (d.get('other_configs') or {}).items():
    print(k, {kk: v.get(kk) for kk in ('value','ms','decode_ms','round_trip_ok','error')}, (v.get('batch_encode') or {}).get('value'))
print('pcie', d.get('pcie_inclusive'))
"
