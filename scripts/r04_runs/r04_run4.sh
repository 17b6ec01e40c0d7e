#!/bin/bash
set -o pipefail
O=gpurun_out/r04; mkdir -p $O
export IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_trace.so
for sh in "L0 64->64 3x3 plain" "L0 64->64 3x3 +stats+pro" "L0up 144->64" "L2up 416->256"; do
  timeout -k 10 120 python scripts/conv_bench.py --only "$sh" --algos 3 --rounds 1 --iters 2 2>&1 | grep -E "wino4 trace|wino4 slots|us " | tail -3
done
