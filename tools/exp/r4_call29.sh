#!/bin/bash
# round 4, late: K3 keeps its window of codes in LDS when a tile takes only some of them (long matches)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/r4_suite_l.log 2>&1; tail -5 $O/r4_suite_l.log
LFX_FUZZ=300 LFX_FUZZ_SEED=11 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | tail -2
timeout 300 python tools/exp/cfg5_run.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print({k:v for k,v in d.items() if 'phase' not in k}); print(d.get('decode_phases_ms'))"
timeout 300 python bench.py --no-traffic --no-cpu-baseline --no-s1 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('S8K', d['value'], d['ms_per_step'], d['phases_ms']); print(d['other_configs'])"
