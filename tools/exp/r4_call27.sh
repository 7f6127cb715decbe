#!/bin/bash
# round 4, late: table build of the block scan (register counts, SWAR rank), 1024-lane symbol kernel for streams of at most one unit per CU
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/r4_suite_k.log 2>&1; tail -8 $O/r4_suite_k.log
timeout 300 python tools/bench_small.py 8192 65536 262144 1048576 4194304 16777216 67108864 > $O/r4_small_j.json 2>/dev/null; python - <<'P'
import json
for l in open('gpurun_out/r4_small_j.json'):
    if l.startswith('{'):
        d=json.loads(l); print(d['bytes'], d['encode_ms'], d['decode_ms'], d['decode_phases_ms'])
P
timeout 300 python bench.py --no-traffic --no-cpu-baseline --no-subs --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('S8K', d['value'], d['ms_per_step'], d['phases_ms']); print(d['schedule_S1'])"
LFX_DEBUG=1 timeout 100 python tools/bench_small.py 1048576 2>&1 >/dev/null | grep "piece 0/" | tail -2 | cut -c1-200
