#!/bin/bash
# round-end artifacts: bench line (default run), kernel stats of the same loop, match-kernel instruction counters, small sizes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python bench.py > $O/r2_bench_default.json 2> $O/r2_bench_default.err; echo "bench rc=$?"; tail -c 600 $O/r2_bench_default.json
cd /tmp
rm -rf $O/kt_r2
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_r2 -- python $R/bench.py --child --steps 5 --warmup 2 > $O/kt_r2.log 2>&1
f=$(find $O/kt_r2 -name "*kernel_stats.csv" | head -1); cp $f $O/kt_r2_kernel_stats.csv; head -5 $f | cut -d, -f1-4
rm -rf $O/pmc_i
timeout 240 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM --output-format csv -d $O/pmc_i -- python $R/bench.py --child --steps 1 --warmup 0 > $O/pmc_i.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_i > $O/r2_pmc_insts.csv 2>/dev/null; grep -E "lz77_match2|materialize2|blk_scan|blk_emit|parse_spec" $O/r2_pmc_insts.csv | head -40
cd $R
timeout 200 python tools/bench_small.py > $O/r2_small.json 2>/dev/null; cat $O/r2_small.json | cut -c1-220
