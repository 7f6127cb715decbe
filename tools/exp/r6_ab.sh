#!/bin/bash
# round 6: A/B of a diagnostic switch on ONE box — the bare bench line (no sub-records) with and without it, twice each
# usage: tools/exp/r6_ab.sh OUTDIR VAR=VALUE [VAR=VALUE ...]
out=$1; shift
mkdir -p "$out"
for rep in 1 2; do
  python bench.py --steps 10 --warmup 3 --no-subs --no-traffic --no-cpu-baseline --no-s1 > "$out/base_$rep.log" 2>&1
  env "$@" python bench.py --steps 10 --warmup 3 --no-subs --no-traffic --no-cpu-baseline --no-s1 > "$out/alt_$rep.log" 2>&1
done
python - "$out" <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.log")):
    for l in open(f):
        if l.startswith('{"metric"'):
            d = json.loads(l)
            p = d["phases_ms"]
            print(f.split("/")[-1], "value", d["value"], "enc", d["encode_GBps"], "dec", d["decode_GBps"],
                  {k: p[k] for k in ("dec:blk_scan", "dec:blk_emit", "dec:find1", "dec:find2", "dec:lz77_copy", "enc:lz77_parse", "enc:lz77_match",
                                     "enc:histogram", "enc:huffman", "enc:pack", "enc:checksum") if k in p})
PY
