#!/bin/bash
# round 4 soak: the nondeterministic failures of lfx_match5's development (stale L1 lines, queue holes, load/store ordering)
# all showed up within a few dozen chunks of random data: 1000 chunks of mixed kinds in one context, then the 256 MiB
# bit-exact comparison three times
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 900 python tools/exp/m5_stress.py 1000 2>&1 | tail -2
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_round2.py -m gpu -q -k "cfg2_256mib" 2>&1 | tail -1; done
