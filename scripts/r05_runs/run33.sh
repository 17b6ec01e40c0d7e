#!/bin/bash
# (run with variants that LACKED -fno-slp-vectorize) conv_wino4_kernel, delay of the first LDS-DMA piece behind every chunk barrier by the wait states of its own s_nop's (same bytes: placement unchanged):
# dhA_B = heavy waves s_nop A / s_nop B instead of 4 / 0, dlA_B = light waves; dsame = the statement rewritten with immediates, 4 / 0 (must equal tree)
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run33; mkdir -p $O
L="tree w4dsame w4dh15_0 w4dh15_15 w4dh9_5 w4dl15_15 w4dl9_5"
for r in 1 2; do for lib in $L; do
  if [ $lib = tree ]; then unset IDIFF_LIB; else export IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$lib.so; fi
  echo -n "== $lib: "; python3 bench.py --no-cpu-baseline --no-train-leg --no-roofline 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'steps/s', d['ms_per_step'], 'ms/step')"
done; done 2>&1 | tee $O/bench.txt
