#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r4_suite_e.log 2>&1; tail -3 $O/r4_suite_e.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-subs --no-cpu-baseline --no-s1 --no-traffic > $O/r4_bench_d.log 2>&1; tail -1 $O/r4_bench_d.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['encode_GBps'], d['decode_GBps']); print({k:round(v,3) for k,v in d['phases_ms'].items()})"
timeout 200 python tools/bench_small.py 65536 1048576 > $O/r4_small_d.json 2>/dev/null; cut -c1-200 $O/r4_small_d.json
