#!/bin/bash
# round 5: the block header's code-length sequence decoded by a wavefront per 64-bit window — parity first, then what it is for:
# one small stream, cfg3 (4096 x 64 KiB), the 256 MiB decode
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_large.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python tools/bench_small.py 65536 1048576 16777216 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['bytes'], 'enc', d['encode_ms'], 'dec', d['decode_ms'], {k: round(v, 3) for k, v in d.get('decode_phases_ms', {}).items()})"
REPS=2 timeout 300 python tools/exp/cfg3_run.py 2>/dev/null | tail -2 | cut -c1-700
timeout 300 python tools/exp/enc_timing.py 268435456 8192 4 2>&1 | grep -E 'rep 3|rror' | sed 's/.*| dec //' | cut -c1-200
