#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
bash tools/exp/r3_quick.sh
echo "== bench"; timeout 900 python bench.py > $O/r3c6_bench.json 2> $O/r3c6_bench.err; echo "bench rc=$?"; python3 - <<PY
import json
try:
    d=json.load(open("$O/r3c6_bench.json"))
    print("value", d["value"], "enc", d["encode_GBps"], "dec", d["decode_GBps"], "ms", d["ms_per_step"])
    print("phases", d["phases_ms"])
    print("roof", {k:v for k,v in d["roofline"].items() if k not in ("traffic_detail","bound_detail")})
    print("bound", d["roofline"].get("bound_detail"))
    print("whole", d["whole_path"]); print("S1", d.get("schedule_S1")); print("other", d.get("other_configs"))
    print("cpu", d["cpu_baseline"]["value"])
    td=d["roofline"].get("traffic_detail") or {}
    print("traffic by kernel", td.get("step_hbm_bytes_by_kernel"))
except Exception as e:
    print("bench parse failed", e); print(open("$O/r3c6_bench.err").read()[-2000:])
PY
