#!/bin/bash
# does hipcc's SLP packing help or hurt the other translation units of the sampling step?  each rebuilt with -fno-slp-vectorize (conv_wino4h: WITH SLP, the
# opposite of its Makefile flag) and linked in place; control first
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run35; mkdir -p $O
L="tree att_noslp x3_noslp smm_noslp elem_noslp igemm_noslp w4h_slp"
for r in 1 2; do for lib in $L; do
  if [ $lib = tree ]; then unset IDIFF_LIB; else export IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$lib.so; fi
  echo -n "== $lib: "; python3 bench.py --no-cpu-baseline --no-train-leg --no-roofline 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'steps/s', d['ms_per_step'], 'ms/step')"
done; done 2>&1 | tee $O/bench.txt
