#!/bin/bash
# training-only translation units rebuilt without SLP packing (it costs conv_wino4_kernel 1.5-2 %): does it cost the weight-gradient / backward kernels too?
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run37; mkdir -p $O
L="tree wg4_ctl wg4_noslp wg_noslp bwd_noslp gemm_noslp"
for r in 1 2; do for lib in $L; do
  if [ $lib = tree ]; then unset IDIFF_LIB; else export IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$lib.so; fi
  echo -n "== $lib: "; python3 bench.py --mode train --batch 32 --steps 5 --warmup 2 --no-roofline 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'it/s', d['ms_per_step'], 'ms')"
done; done 2>&1 | tee $O/bench.txt
