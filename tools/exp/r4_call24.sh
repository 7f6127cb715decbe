#!/bin/bash
# round 4, late: Huffman stage rewrite, tree-less checksum combine, batched window-chain loads, wide K3 for few-block streams
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/r4_suite_i.log 2>&1; tail -15 $O/r4_suite_i.log
LFX_DEBUG=1 timeout 120 python tools/bench_small.py 1048576 2>&1 >/dev/null | grep "huffman block 0" | tail -1
timeout 300 python tools/bench_small.py 65536 262144 1048576 4194304 16777216 67108864 > $O/r4_small_h.json 2>/dev/null; python - <<'P'
import json
for l in open('gpurun_out/r4_small_h.json'):
    if l.startswith('{'):
        d=json.loads(l); print(d['bytes'], d['encode_ms'], d['decode_ms'], {k:v for k,v in d['encode_phases_ms'].items() if k in('huffman','lz77_match')}, d['decode_phases_ms'])
P
timeout 600 python bench.py --no-traffic --no-cpu-baseline > $O/r4_bench_i.json 2> $O/r4_bench_i.err; echo "bench rc=$?"; python - <<'P'
import json
l=[x for x in open('gpurun_out/r4_bench_i.json') if x.startswith('{')][-1]
d=json.loads(l); print(d['value'], d['ms_per_step'], d['encode_GBps'], d['decode_GBps']); print(d['phases_ms']); print(d['schedule_S1']); print({k:(v.get('value'),v.get('batch_encode',{}).get('value')) for k,v in d['other_configs'].items()})
P
timeout 300 python bench.py --schedule S1 --no-traffic --no-cpu-baseline --no-s1 --no-subs --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('S1', d['value'], d['ms_per_step'], d['phases_ms'])"
