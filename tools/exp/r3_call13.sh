#!/bin/bash
# the N>1 bench path on a one-GPU box: sharded (RCCL) branch at world_size 1, and 3 ranks sharing GPU 0 over gloo
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
LFX_BENCH_FORCE_SHARDED=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > $O/r3_bench_force_sharded.json 2> $O/r3_bench_force_sharded.err; echo "forced rc=$?"; cut -c1-900 $O/r3_bench_force_sharded.json; tail -3 $O/r3_bench_force_sharded.err
LFX_BENCH_ONE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 3 --steps 2 --warmup 1 > $O/r3_bench_one_gpu_3ranks.json 2> $O/r3_bench_one_gpu_3ranks.err; echo "3 ranks rc=$?"; cut -c1-900 $O/r3_bench_one_gpu_3ranks.json; tail -5 $O/r3_bench_one_gpu_3ranks.err
