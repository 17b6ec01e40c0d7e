#!/bin/bash
set -o pipefail
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -q -x -k "winograd4 or groupnorm" > $O/t7_ops.log 2>&1 || { grep -E "^FAILED|passed|failed|Error" $O/t7_ops.log | tail; exit 1; }
tail -1 $O/t7_ops.log
timeout -k 10 300 python scripts/conv_bench.py --only 3x3 --algos 3 2>&1 | grep " us " | tee $O/cb7.log
bash scripts/r04_bench.sh spec2reg
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/t7_full.log 2>&1 || { echo "full FAILED"; tail -40 $O/t7_full.log; exit 1; }
tail -2 $O/t7_full.log
