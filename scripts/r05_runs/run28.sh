#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run28; mkdir -p $O
for lib in w4p3 w4p2; do
  IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$lib.so timeout -k 10 300 python3 -m pytest tests/test_ops_gpu.py -x -q -k "winograd4" 2>&1 | tail -1 || exit 1
done | tee $O/tests.txt
for lib in tree w4old_inplace w4p3 w4p2 tree w4old_inplace w4p3 w4p2; do
  if [ $lib = tree ]; then unset IDIFF_LIB; else export IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$lib.so; fi
  echo -n "== $lib: "; python3 bench.py --no-cpu-baseline --no-train-leg --no-roofline 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'steps/s', d['ms_per_step'], 'ms/step')"
done 2>&1 | tee $O/bench.txt
for lib in tree w4p2; do
  if [ $lib = tree ]; then unset IDIFF_LIB; else export IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$lib.so; fi
  echo "== $lib"; python3 scripts/conv_bench.py --only "pro" --rounds 4 --iters 5 --algos 3 2>&1 | grep "algo 3"
done 2>&1 | tee $O/ab.txt
