#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q 2>&1 | tail -2
timeout 400 python bench.py --steps 8 --warmup 2 --no-subs --no-cpu-baseline --no-s1 > $O/r4_bench_e.log 2>&1; tail -1 $O/r4_bench_e.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['encode_GBps'], d['decode_GBps']); print({k:round(v,3) for k,v in d['phases_ms'].items()}); print(d['whole_path']); print(d['roofline']['traffic_detail']['step_hbm_bytes_by_kernel'])"
