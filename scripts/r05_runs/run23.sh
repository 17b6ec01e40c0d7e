#!/bin/bash
# per-item cycle table of conv_wino4_kernel (trace builds, with the Makefile's -fno-slp-vectorize: w4trace0 = -DIDIFF_WINO_TRACE, s_memtime at the four phase
# boundaries of wave 0 only; w4trace = + -DIDIFF_WINO_SLOTS,
# per-wave work / barrier-wait cycles per chunk), on the layer shapes of the c2 step
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run23; mkdir -p $O
for lib in w4trace0 w4trace; do
export IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$lib.so; echo "== $lib"
for s in "L0 64->64 3x3 plain" "L0 64->64 3x3 +stats" "L0 64->64 3x3 +stats+pro" "L0up 144->64" "L1 64->64" "L2 128->128" "L2up 416->256" "up 128->64"; do
  python3 scripts/conv_bench.py --only "$s" --rounds 1 --iters 1 --algos 3 2>&1 | grep -E "wino4 trace|wino4 slots|us " | sort | uniq -c | sort -rn | head -4
done; done 2>&1 | tee $O/w4_item_cycles.txt
