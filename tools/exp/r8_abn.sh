#!/bin/bash
# GPU box: parity tests on the product build, then the bare bench line and cfg5 for the product and every variant library given
# usage: tools/exp/r8_abn.sh OUT "pytest -k expression" tag1 tag2 ...   (libflate_amd/liblfx_<tag>.so)
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; kexpr=$2; shift; shift
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "$kexpr" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
libs=""; for t in "$@"; do libs="$libs libflate_amd/liblfx_$t.so"; done
bash tools/exp/r6_so_ab.sh $out $libs | grep -v pytest | python -c "
import sys,re
for l in sys.stdin:
    m=re.match(r'(\S+) value (\S+) enc (\S+) dec (\S+) .*lz77_parse.: ([\d.]+)', l)
    print(m.groups() if m else l.strip())
"
for v in "" "$@"; do
  so=libflate_amd/liblfx${v:+_$v}.so
  LFX_SO=$PWD/$so timeout 300 python tools/exp/cfg5_run.py 1073741824 3 2>&1 | tail -1 > $out/cfg5_$v.json
  python - $out/cfg5_$v.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1], d["encode_GBps"], d["encode_phases_ms"]["lz77_parse"], d["encode_phases_ms"]["lz77_match"], d["round_trip_ok"])
PY
done
