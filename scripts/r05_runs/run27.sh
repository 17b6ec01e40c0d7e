#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run27; mkdir -p $O
for r in 1 2; do
for lib in tree ord_w4first ord_w4last w4old_inplace w4prio; do
  if [ $lib = tree ]; then unset IDIFF_LIB; else export IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$lib.so; fi
  echo "== $lib"; python3 scripts/conv_bench.py --only "3x3" --rounds 4 --iters 5 --algos 3 2>&1 | grep -v amdgpu.ids | grep "algo 3" | grep -v final | awk '{printf "%s %s %s | ", $1, $2, $(NF-7)} END {print ""}'
done; done 2>&1 | tee $O/ab.txt
for lib in tree ord_w4first ord_w4last w4old_inplace w4prio tree ord_w4first ord_w4last w4old_inplace w4prio; do
  if [ $lib = tree ]; then unset IDIFF_LIB; else export IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$lib.so; fi
  echo -n "== $lib: "; python3 bench.py --no-cpu-baseline --no-train-leg --no-roofline 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'steps/s', d['ms_per_step'], 'ms/step')"
done 2>&1 | tee $O/bench.txt
