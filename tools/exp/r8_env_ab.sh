#!/bin/bash
# GPU box: parity tests, then the bare bench line and cfg5 with and without an environment switch
# usage: tools/exp/r8_env_ab.sh OUT "pytest -k expression" VAR=1
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; kexpr=$2; sw=$3
mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -x -q -k "$kexpr" > $out/pytest.log 2>&1; tail -3 $out/pytest.log; grep -c "corrupted\|Aborted" $out/pytest.log
for rep in 1 2; do
  for v in "" "$sw"; do
    env $v python bench.py --steps 10 --warmup 3 --no-subs --no-traffic --no-cpu-baseline --no-s1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); p=d['phases_ms']; print('[$v]', d['value'], d['encode_GBps'], d['decode_GBps'], 'parse', p['enc:lz77_parse'], 'match', p['enc:lz77_match'])"
  done
done
for v in "" "$sw"; do
  env $v timeout 300 python tools/exp/cfg5_run.py 1073741824 3 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg5 [$v]', d['encode_GBps'], d['encode_phases_ms']['lz77_parse'], d['round_trip_ok'])"
done
