#!/bin/bash
# round 6: the bare bench line with development builds of the library (LFX_SO), each twice, beside the product build
# usage: tools/exp/r6_so_ab.sh OUTDIR lib1.so [lib2.so ...]     (paths relative to the repository root)
out=$1; shift
mkdir -p "$out"
for rep in 1 2; do
  python bench.py --steps 10 --warmup 3 --no-subs --no-traffic --no-cpu-baseline --no-s1 > "$out/product_$rep.log" 2>&1
  for so in "$@"; do
    LFX_SO=$GRAFT_REPO_ROOT/$so python bench.py --steps 10 --warmup 3 --no-subs --no-traffic --no-cpu-baseline --no-s1 > "$out/$(basename $so .so)_$rep.log" 2>&1
  done
done
python - "$out" <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.log")):
    ok = False
    for l in open(f):
        if l.startswith('{"metric"'):
            d = json.loads(l); p = d["phases_ms"]; ok = True
            print(f.split("/")[-1], "value", d["value"], "enc", d["encode_GBps"], "dec", d["decode_GBps"],
                  {k: p[k] for k in ("enc:lz77_parse", "enc:lz77_match", "enc:histogram", "enc:huffman", "enc:checksum", "dec:find1", "dec:find2", "dec:blk_scan", "dec:blk_emit", "dec:lz77_copy")})
    if not ok:
        print(f.split("/")[-1], "NO LINE:", open(f).read()[-300:].replace("\n", " | "))
PY
