#!/bin/bash
# (run with variants that LACKED -fno-slp-vectorize) stagger probe of conv_wino4_kernel at MATCHED code placement: four s_nop N behind every chunk barrier on the light (sl) / heavy (sh) waves,
# N = 0 / 7 / 15 (4 / 32 / 64 wait states: the same bytes, another delay)
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run32; mkdir -p $O
L="tree w4sl0 w4sl7 w4sl15 w4sh0 w4sh7 w4sh15"
for r in 1 2; do for lib in $L; do
  if [ $lib = tree ]; then unset IDIFF_LIB; else export IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$lib.so; fi
  echo -n "== $lib: "; python3 bench.py --no-cpu-baseline --no-train-leg --no-roofline 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'steps/s', d['ms_per_step'], 'ms/step')"
done; done 2>&1 | tee $O/bench.txt
