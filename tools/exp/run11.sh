#!/bin/bash
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "cfg4 or tiny" 2>&1 | tail -5
  echo "=== default bench"
  timeout 900 python bench.py 2> gpurun_out/r2_bench_default.err | tee gpurun_out/r2_bench_default.json | cut -c1-3000
  tail -3 gpurun_out/r2_bench_default.err
  echo "=== nccl world 1, sharded path"
  LFX_BENCH_FORCE_SHARDED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -3 | cut -c1-1500
  echo "=== gloo world 2 on one GPU"
  LFX_BENCH_ONE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 2 --warmup 1 --bytes 67108864 2>&1 | tail -3 | cut -c1-1500
) > gpurun_out/r2_bench.log 2>&1
cat gpurun_out/r2_bench.log
