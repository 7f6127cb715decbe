#!/bin/bash
# round-end artifacts (round 6): full GPU suite, bench line (default run), kernel stats of the same loop, SQ counters per kernel,
# cfg5 / cfg3 kernel stats, small sizes, S1 phases, fuzz soak, the N-GPU drivers on one GPU (with overlap_ms), the host-memory
# surface (pcie_inclusive, the io::copy protocol at several batch sizes), the single-pass A/B
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -q -m gpu > $O/r6_suite.log 2>&1; tail -4 $O/r6_suite.log | grep -v amdgpu > $O/r06_gpu_suite_tail.txt; cat $O/r06_gpu_suite_tail.txt
timeout 1200 python bench.py > $O/r06_bench_default.json 2> $O/r06_bench_default.err; echo "bench rc=$?"; cut -c1-400 $O/r06_bench_default.json
cd /tmp
rm -rf /tmp/kt_r5
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_r5 -- python $R/bench.py --child --steps 5 --warmup 2 > /tmp/kt_r5.log 2>&1
python $R/tools/prof_summary.py /tmp/kt_r5 "rocprofv3 --kernel-trace --stats -- python bench.py --child --steps 5 --warmup 2 (256 MiB TEXT, S8K, gzip encode + decode)" 2>/dev/null | grep -v "at::native\|elementwise" > $O/r06_kernel_stats.csv; head -8 $O/r06_kernel_stats.csv | cut -c1-150
for pass in 1 2 3; do
  if [ $pass = 1 ]; then C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS";
  elif [ $pass = 2 ]; then C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_BUSY_CYCLES";
  else C="SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE"; fi
  rm -rf /tmp/pmc_r6_$pass
  timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_r6_$pass -- python $R/bench.py --child --steps 1 --warmup 0 > /tmp/pmc_r6_$pass.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pmc_r6_$pass 2>/dev/null | grep -v "at::native\|elementwise" > /tmp/r6_pmc_$pass.csv
done
cat /tmp/r6_pmc_1.csv /tmp/r6_pmc_2.csv /tmp/r6_pmc_3.csv > $O/r06_pmc_counters.csv; wc -l $O/r06_pmc_counters.csv
cd $R
# cfg5 (1 GiB LOWENT): kernel stats of THIS build
cd /tmp; rm -rf /tmp/kt_r5c5
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt_r5c5 -- python $R/tools/exp/cfg5_run.py > $O/r06_cfg5_phases.json 2>/dev/null
python $R/tools/prof_summary.py /tmp/kt_r5c5 "rocprofv3 --kernel-trace --stats -- python tools/exp/cfg5_run.py (zlib encode + decode of 1 GiB LOWENT, 8192-byte writes)" 2>/dev/null | grep -v "at::native\|elementwise" > $O/r06_cfg5_kernel_stats.csv; head -6 $O/r06_cfg5_kernel_stats.csv | cut -c1-150
# cfg3 (4096 x 64 KiB): kernel stats of one batch encode + four batch decodes
cd /tmp; rm -rf /tmp/kt_c3
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt_c3 -- python $R/tools/exp/cfg3_run.py 2>/dev/null | tail -2 > $O/r06_cfg3_phases.json
python $R/tools/prof_summary.py /tmp/kt_c3 "rocprofv3 --kernel-trace --stats -- python tools/exp/cfg3_run.py (4096 x 64 KiB zlib streams: one batch encode of 2048, four batch decodes of 4096)" 2>/dev/null | grep -v "at::native\|elementwise" > $O/r06_cfg3_kernel_stats.csv; head -5 $O/r06_cfg3_kernel_stats.csv | cut -c1-150
cd $R
timeout 300 python tools/bench_small.py 8192 65536 262144 1048576 4194304 16777216 33554432 67108864 100663296 134217728 > $O/r06_small_sizes.json 2>/dev/null; cut -c1-300 $O/r06_small_sizes.json
timeout 300 python bench.py --schedule S1 --no-traffic --no-cpu-baseline --no-s1 --no-subs --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/r06_bench_s1.json; cut -c1-200 $O/r06_bench_s1.json
LFX_FUZZ=500 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | tail -2 | grep -v amdgpu > $O/r06_fuzz.txt; cat $O/r06_fuzz.txt
timeout 400 python tools/exp/m5_stress.py 1000 2>&1 | tail -1 >> $O/r06_fuzz.txt; tail -1 $O/r06_fuzz.txt
LFX_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-subs --no-cpu-baseline --no-s1 --no-traffic 2>/dev/null | tail -1 > $O/r06_bench_force_sharded_world1.txt; cut -c1-160 $O/r06_bench_force_sharded_world1.txt
LFX_BENCH_ONE_GPU=1 LFX_BENCH_CFG4_BYTES=134217728 timeout 900 python bench.py --gpus 3 --steps 3 --warmup 1 --bytes 67108864 --no-cpu-baseline --no-s1 --no-traffic 2>/dev/null | grep "^{" | tail -1 > $O/r06_bench_one_gpu_3ranks.txt; cut -c1-160 $O/r06_bench_one_gpu_3ranks.txt
LFX_BENCH_ONE_GPU=1 timeout 900 python bench.py --gpus 3 --steps 3 --warmup 1 --scaling strong --no-subs --no-cpu-baseline --no-s1 --no-traffic 2>/dev/null | grep "^{" | tail -1 > $O/r06_bench_one_gpu_3ranks_strong.txt; cut -c1-160 $O/r06_bench_one_gpu_3ranks_strong.txt
python tools/exp/r6_hostio.py 2>/dev/null | grep "^{" > $O/r06_host_surface.jsonl; cut -c1-200 $O/r06_host_surface.jsonl
tools/exp/r6_ab.sh /tmp/r6_ab LFX_TWO_PASS=1 > $O/r06_ab_single_pass.txt 2>&1; cat $O/r06_ab_single_pass.txt | cut -c1-300
