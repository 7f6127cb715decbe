#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out; mkdir -p $O
cd $R
echo "== diag"; timeout 600 python tools/exp/r3_diag.py 2>&1 | grep -vE "^==== env|amdgpu.ids" | head -16
echo "== round-3 tests"; timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu > $O/r3c4_t3.log 2>&1; tail -5 $O/r3c4_t3.log
echo "== timing S8K"; LFX_DEBUG=1 timeout 300 python tools/exp/enc_timing.py 268435456 8192 3 > $O/r3c4_timing.log 2>&1; grep -E "rep 2|crc ok|Error|error" $O/r3c4_timing.log | cut -c1-700; grep "match3 wave" $O/r3c4_timing.log | tail -16 | head -2
echo "== timing S1"; timeout 300 python tools/exp/enc_timing.py 268435456 0 2 2>&1 | grep -E "rep 1|crc ok|rror" | cut -c1-700
echo "== oracle 64 MiB"; timeout 300 python tools/exp/enc_timing.py 67108864 8192 1 2>&1 | grep -E "equal|rror"
echo "== full gpu suite"; timeout 1500 python -m pytest tests -q -m gpu -x > $O/r3c4_suite.log 2>&1; tail -8 $O/r3c4_suite.log
cd /tmp
rm -rf $O/kt_r3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_r3 -- python $R/bench.py --child --steps 5 --warmup 2 > $O/kt_r3.log 2>&1
f=$(find $O/kt_r3 -name "*kernel_stats.csv" | head -1); cp $f $O/r3c4_kernel_stats.csv; rm -rf $O/kt_r3; python3 - <<PY
import csv
for r in list(csv.reader(open("$O/r3c4_kernel_stats.csv")))[:22]:
    print(r[0][:60], r[1], r[3])
PY
