#!/bin/bash
# round 4, late: short warm-up for small pieces, small streams start without the finder from 4 KiB on, single-piece blocks take the direct path
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/r4_suite_j.log 2>&1; tail -8 $O/r4_suite_j.log
timeout 300 python tools/bench_small.py 8192 32768 65536 131072 262144 1048576 4194304 16777216 67108864 > $O/r4_small_i.json 2>/dev/null; python - <<'P'
import json
for l in open('gpurun_out/r4_small_i.json'):
    if l.startswith('{'):
        d=json.loads(l); print(d['bytes'], d['encode_ms'], d['decode_ms'], d['decode_phases_ms'])
P
timeout 300 python bench.py --no-traffic --no-cpu-baseline --no-s1 --no-subs --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('S8K', d['value'], d['ms_per_step'], d['phases_ms'])"
