#!/bin/bash
# conv_wino4_kernel variants REBUILT WITH THE MAKEFILE'S PER-OBJECT FLAGS (-fno-slp-vectorize; run24..run33 lacked it): control = the tree's source through
# the variant builder (w4ctl), the 4 + 5 split of the prologue instantiation (w4old), 5 + 4 (w4u5h), padding / alignment (w4pad*, w4al*), same-bytes delay
# of the first DMA piece behind the chunk barrier (w4d*), s_setprio 1 on waves 4-7 (w4prio)
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run34; mkdir -p $O
L="tree w4ctl w4old w4u5h w4padh1 w4padl1 w4al0 w4al3 w4dsame w4dh15_15 w4dh9_5 w4dl15_15 w4prio"
for r in 1 2; do for lib in $L; do
  if [ $lib = tree ]; then unset IDIFF_LIB; else export IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$lib.so; fi
  echo -n "== $lib: "; python3 bench.py --no-cpu-baseline --no-train-leg --no-roofline 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'steps/s', d['ms_per_step'], 'ms/step')"
done; done 2>&1 | tee $O/bench.txt
