#!/bin/bash
set -o pipefail
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -q -x -k "winograd4 or groupnorm" > $O/t5_ops.log 2>&1 || { grep -E "^FAILED|passed|failed" $O/t5_ops.log | tail; exit 1; }
tail -1 $O/t5_ops.log
timeout -k 10 300 python scripts/conv_bench.py --only 3x3 --algos 3,4 2>&1 | grep " us " | tee $O/cb5.log
export IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_trace.so
for sh in "L0 64->64 3x3 plain" "L0 64->64 3x3 +stats+pro" "L2up 416->256"; do
  timeout -k 10 120 python scripts/conv_bench.py --only "$sh" --algos 3 --rounds 1 --iters 2 2>&1 | grep -E "wino4 trace|wino4 slots|us " | tail -3
done
