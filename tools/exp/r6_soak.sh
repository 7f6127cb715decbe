#!/bin/bash
# round 6 soak on the final build: differential fuzz (2500 trials), foreign-stream fuzz (150), randomized suite (LFX_FUZZ=800),
# 1500 stress chunks through the candidate stage, the 256 MiB bit-exact comparison twice
cd $GRAFT_REPO_ROOT; O=gpurun_out
{
echo "# tools/exp/r6_soak.sh, final round-6 build"
LFX_FUZZ_TRIALS=2500 timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k differential_random 2>&1 | tail -1
LFX_FOREIGN_TRIALS=150 timeout 400 python -m pytest tests/test_gpu_large.py -x -q -m gpu -k "foreign_streams_fuzz" 2>&1 | tail -1
LFX_FUZZ=800 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | tail -1
timeout 600 python tools/exp/m5_stress.py 1500 2>&1 | tail -1
for i in 1 2; do timeout 600 python -m pytest tests/test_gpu_round2.py -m gpu -q -k "cfg2_256mib" 2>&1 | tail -1; done
# round 6: the decode paths side by side (single pass / tight regions = fallback / two passes) on the randomized suite, and the host-memory tests
for v in LFX_TWO_PASS LFX_STORE_TIGHT LFX_HIST_SEPARATE LFX_NO_PIN_SLOTS; do env $v=1 LFX_FUZZ=300 timeout 600 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_large.py -q -m gpu 2>&1 | tail -1; done
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_round6.py -m gpu -q 2>&1 | tail -1; done
} | grep -v amdgpu | tee $O/r06_soak.txt
