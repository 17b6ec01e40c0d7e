#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run22; mkdir -p $O
timeout -k 10 800 python3 -m pytest tests/test_sampling_gpu.py -x -q -s -k "flash or hierarch" 2>&1 | grep -v amdgpu.ids | tail -25 | tee $O/tests.txt
