#!/bin/bash
# GPU box: parity tests of the encode path on the product build, then the bare bench line product vs liblfx_a.so (baseline), cfg5 on both
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-r8b}; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "${2:-encode or parse or cfg2 or lz77}" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
bash tools/exp/r6_so_ab.sh $out libflate_amd/liblfx_a.so
for v in "" _a; do
  LFX_SO=$PWD/libflate_amd/liblfx$v.so timeout 300 python tools/exp/cfg5_run.py 1073741824 3 2>&1 | tail -1 > $out/cfg5$v.json
  python - $out/cfg5$v.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1], d["encode_GBps"], d["encode_phases_ms"]["lz77_parse"], d["encode_phases_ms"]["lz77_match"], d["round_trip_ok"])
PY
done
