#!/bin/bash
# code-placement scan of conv_wino4_kernel: the main chunk loop's head anchored with .p2align 6, then K x 4 bytes of s_nop, K = 0..15
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run30; mkdir -p $O
L="tree w4al0 w4al1 w4al2 w4al3 w4al4 w4al5 w4al6 w4al7 w4al8 w4al9 w4al10 w4al11 w4al12 w4al13 w4al14 w4al15 tree"
for lib in $L; do
  if [ $lib = tree ]; then unset IDIFF_LIB; else export IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$lib.so; fi
  echo -n "== $lib: "; python3 bench.py --no-cpu-baseline --no-train-leg --no-roofline 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'steps/s', d['ms_per_step'], 'ms/step')"
done 2>&1 | tee $O/bench.txt
