#!/bin/bash
# soak: many more fuzz trials than the default suite runs (the materialise kernels' in-place pointer jumping relies on
# the LDS keeping a wavefront's order — any violation shows up as a wrong byte / CRC error)
cd $GRAFT_REPO_ROOT
LFX_FUZZ_TRIALS=2500 timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k differential_random 2>&1 | tail -2
LFX_FOREIGN_TRIALS=150 timeout 400 python -m pytest tests/test_gpu_large.py -x -q -m gpu -k "foreign_streams_fuzz" 2>&1 | tail -2
