#!/bin/bash
# rebuild liblfx.so; non-zero exit on any compile error (so that `build.sh && gpurun …` never ships a stale library)
cd "$(dirname "$0")/../.." || exit 1
out=$(python -c "from libflate_amd import build; build.build()" 2>&1); rc=$?
echo "$out" | grep -E "error|warning" | head -20
[ $rc -eq 0 ] || { echo "BUILD FAILED"; exit 1; }
