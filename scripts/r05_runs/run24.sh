#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run25; mkdir -p $O
V=$PWD/instancediff_amd/variants/libidiff_w4u36.so
IDIFF_LIB=$V timeout -k 10 400 python3 -m pytest tests/test_ops_gpu.py -x -q -k "winograd4" 2>&1 | tail -2 | tee $O/tests.txt || exit 1
for r in 1 2; do
for lib in base w4u36; do
  if [ $lib = base ]; then unset IDIFF_LIB; else export IDIFF_LIB=$V; fi
  echo "== $lib"; python3 scripts/conv_bench.py --only "3x3" --rounds 5 --iters 5 --algos 3 2>&1 | grep -v amdgpu.ids | grep "algo 3"
done; done 2>&1 | tee $O/ab.txt
for lib in base w4u36 base w4u36; do
  if [ $lib = base ]; then unset IDIFF_LIB; else export IDIFF_LIB=$V; fi
  echo "== $lib"; python3 bench.py --no-cpu-baseline --no-train-leg --no-roofline 2>/dev/null | cut -c1-170
done 2>&1 | tee $O/bench.txt
