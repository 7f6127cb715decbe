#!/bin/bash
# round 3, first GPU call: does the lazy-length encode path work, what does it cost, what bounds its kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out; mkdir -p $O
cd $R
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== round-3 tests"; timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu 2>&1 | tail -15
echo "== timing S8K"; LFX_DEBUG=1 timeout 300 python tools/exp/enc_timing.py 268435456 8192 3 > $O/r3c1_timing.log 2>&1; grep -E "rep 2|crc ok|Error|error" $O/r3c1_timing.log | cut -c1-700; grep "match3 wave" $O/r3c1_timing.log | tail -16 | head -3
echo "== timing S1"; timeout 300 python tools/exp/enc_timing.py 268435456 0 2 2>&1 | grep -E "rep 1|crc ok|rror" | cut -c1-700
echo "== oracle 64 MiB"; timeout 300 python tools/exp/enc_timing.py 67108864 8192 1 2>&1 | grep -E "equal|rror"
echo "== lowent 256 MiB"; timeout 300 python - <<'PY' 2>&1 | tail -4
import sys, time
sys.path[:0] = ["tools", "oracle"]
import ctypes as C, numpy as np, torch, synth, libflate_amd
from libflate_amd import _ffi
ctx = libflate_amd.Context(0); ctx.enable_timing(True)
n = 256 << 20
data = synth.lowent(n); d_in = torch.from_numpy(data).cuda()
opts, sched = _ffi.make_opts(), _ffi.make_schedule(8192)
bound = _ffi.lib().lfx_encode_bound(n, C.byref(opts), C.byref(sched)) & ~3
d_out = torch.empty(bound, dtype=torch.uint8, device="cuda"); d_dec = torch.empty(n, dtype=torch.uint8, device="cuda")
for r in range(2):
    m = ctx.encode_device(_ffi.ZLIB, d_in.data_ptr(), n, d_out.data_ptr(), bound, opts, sched); te = ctx.last_timing()
    rc, ol, used, msg = ctx.decode_device(_ffi.ZLIB, d_out.data_ptr(), m, d_dec.data_ptr(), n); td = ctx.last_timing()
    print("lowent rep", r, "m", m, "enc %.3f dec %.3f" % (te["total_ms"], td["total_ms"]), " ".join("%s=%.3f" % kv for kv in te["phases"]))
print("roundtrip", rc, torch.equal(d_dec, d_in))
PY
echo "== full gpu suite"; timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12
cd /tmp
for pass in 1 2; do
  if [ $pass = 1 ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA";
  else C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; fi
  rm -rf $O/pmc_r3_$pass
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/pmc_r3_$pass -- python $R/bench.py --child --steps 1 --warmup 1 > $O/pmc_r3_$pass.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc_r3_$pass > $O/r3c1_pmc_$pass.csv 2>/dev/null
  rm -rf $O/pmc_r3_$pass
done
grep -E "lz77_match3|parse_walk|blk_scan|blk_emit|materialize2|find_blocks" $O/r3c1_pmc_1.csv $O/r3c1_pmc_2.csv | cut -d: -f2 | head -100
rm -rf $O/kt_r3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_r3 -- python $R/bench.py --child --steps 5 --warmup 2 > $O/kt_r3.log 2>&1
f=$(find $O/kt_r3 -name "*kernel_stats.csv" | head -1); cp $f $O/r3c1_kernel_stats.csv; rm -rf $O/kt_r3; head -24 $O/r3c1_kernel_stats.csv | cut -d, -f1-4
cd $R
echo "== bench"; timeout 900 python bench.py > $O/r3c1_bench.json 2> $O/r3c1_bench.err; echo "bench rc=$?"; cut -c1-1500 $O/r3c1_bench.json
