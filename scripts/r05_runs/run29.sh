#!/bin/bash
# (run with variants that LACKED -fno-slp-vectorize: see profiles/r05/x_wino4_variants.txt) code-placement probe of conv_wino4_kernel: the tree's kernel with 4 / 8 bytes of s_nop in front of the heavy / light wave class's body
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run29; mkdir -p $O
L="tree w4pad_h1 w4pad_h2 w4pad_l1 w4pad_h1l1"
for r in 1 2; do for lib in $L; do
  if [ $lib = tree ]; then unset IDIFF_LIB; else export IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$lib.so; fi
  echo -n "== $lib: "; python3 bench.py --no-cpu-baseline --no-train-leg --no-roofline 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'steps/s', d['ms_per_step'], 'ms/step')"
done; done 2>&1 | tee $O/bench.txt
for lib in $L; do
  if [ $lib = tree ]; then unset IDIFF_LIB; else export IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$lib.so; fi
  echo "== $lib"; python3 scripts/conv_bench.py --only "3x3" --rounds 3 --iters 5 --algos 3 2>&1 | grep "algo 3" | grep -v "final\|L3" | cut -c1-62
done 2>&1 | tee $O/ab.txt
