#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out
LFX_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-subs --no-cpu-baseline --no-s1 --no-traffic > $O/r4_bench_sharded1.log 2>&1; tail -1 $O/r4_bench_sharded1.log | cut -c1-1200
timeout 300 python bench.py --steps 10 --warmup 3 --no-subs --no-cpu-baseline --no-s1 --no-traffic > $O/r4_bench_c.log 2>&1; tail -1 $O/r4_bench_c.log | cut -c1-1200
LFX_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus 3 --steps 3 --warmup 1 --bytes 67108864 --no-subs --no-cpu-baseline --no-s1 --no-traffic > $O/r4_bench_3ranks.log 2>&1; tail -1 $O/r4_bench_3ranks.log | cut -c1-600
