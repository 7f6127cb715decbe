#!/bin/bash
# round-end artifacts: full GPU suite, bench line (default run), kernel stats of the same loop, SQ counters per kernel, small sizes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -q -m gpu > $O/r4_suite.log 2>&1; tail -4 $O/r4_suite.log
timeout 900 python bench.py > $O/r4_bench_default.json 2> $O/r4_bench_default.err; echo "bench rc=$?"; cut -c1-400 $O/r4_bench_default.json
cd /tmp
rm -rf $O/kt_r4
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_r4 -- python $R/bench.py --child --steps 5 --warmup 2 > $O/kt_r4.log 2>&1
f=$(find $O/kt_r4 -name "*kernel_stats.csv" | head -1); cp $f $O/r4_kernel_stats.csv; rm -rf $O/kt_r4; head -4 $O/r4_kernel_stats.csv | cut -c1-200
for pass in 1 2; do
  if [ $pass = 1 ]; then C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS";
  else C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM"; fi
  rm -rf $O/pmc_r4_$pass
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/pmc_r4_$pass -- python $R/bench.py --child --steps 1 --warmup 0 > $O/pmc_r4_$pass.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc_r4_$pass > $O/r4_pmc_$pass.csv 2>/dev/null
  rm -rf $O/pmc_r4_$pass
done
cat $O/r4_pmc_1.csv $O/r4_pmc_2.csv | grep -v "at::native" > $O/r4_pmc_counters.csv; wc -l $O/r4_pmc_counters.csv
# in-flight levels: SQ_INST_LEVEL_{VMEM,LDS} / SQ_INSTS_{VMEM,LDS} = average latency of vector-memory / LDS instructions (in LDS-latency units)
rm -rf $O/pmc_r4_lat
timeout 300 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_INSTS_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_r4_lat -- python $R/bench.py --child --steps 1 --warmup 0 > $O/pmc_r4_lat.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_r4_lat 2>/dev/null | grep -v "at::native" > $O/r4_pmc_latency.csv; rm -rf $O/pmc_r4_lat; wc -l $O/r4_pmc_latency.csv
cd $R
timeout 300 python tools/bench_small.py 8192 65536 262144 1048576 4194304 16777216 33554432 67108864 100663296 134217728 > $O/r4_small.json 2>/dev/null; cut -c1-300 $O/r4_small.json
timeout 300 python bench.py --schedule S1 --no-traffic --no-cpu-baseline --no-s1 --no-subs --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/r4_bench_s1.json; cut -c1-200 $O/r4_bench_s1.json
LFX_DEBUG=1 timeout 120 python tools/bench_small.py 1048576 2>&1 >/dev/null | grep 'huffman block 0' | tail -1 > $O/r4_huffman_stamps.txt; cat $O/r4_huffman_stamps.txt
timeout 300 python tools/exp/cfg5_run.py > $O/r4_cfg5.json 2>/dev/null; cut -c1-200 $O/r4_cfg5.json
LFX_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-subs --no-cpu-baseline --no-s1 --no-traffic 2>/dev/null | tail -1 > $O/r4_bench_force_sharded_world1.txt; cut -c1-160 $O/r4_bench_force_sharded_world1.txt
LFX_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus 3 --steps 3 --warmup 1 --bytes 67108864 --no-subs --no-cpu-baseline --no-s1 --no-traffic 2>/dev/null | tail -1 > $O/r4_bench_one_gpu_3ranks.txt; cut -c1-160 $O/r4_bench_one_gpu_3ranks.txt
