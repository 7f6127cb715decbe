#!/bin/bash
# round 5: the N-GPU drivers of the library (lfx_sharded_*) on a one-GPU box — the C shim at world size 1, three ranks sharing
# GPU 0 over gloo (weak and strong scaling, the cfg4 sub-record at a small size), the RCCL branch at world size 1
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round4.py tests/test_gpu_parity.py -x -q -m gpu -k "shim or shard or batch or cfg4 or virtual" 2>&1 | tail -3
LFX_BENCH_ONE_GPU=1 LFX_BENCH_CFG4_BYTES=67108864 timeout 900 python bench.py --gpus 3 --steps 3 --warmup 1 --bytes 33554432 --no-cpu-baseline --no-traffic 2>&1 | tail -1 | tee $O/r5_bench_one_gpu_3ranks_weak.txt | cut -c1-900
LFX_BENCH_ONE_GPU=1 timeout 900 python bench.py --gpus 3 --steps 3 --warmup 1 --scaling strong --bytes 100663296 --no-cpu-baseline --no-traffic --no-subs 2>&1 | tail -1 | tee $O/r5_bench_one_gpu_3ranks_strong.txt | cut -c1-600
LFX_BENCH_FORCE_SHARDED=1 timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-subs 2>&1 | tail -1 | tee $O/r5_bench_force_sharded_world1.txt | cut -c1-600
