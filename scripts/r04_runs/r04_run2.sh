#!/bin/bash
# round 4, GPU run 2: the all-DMA streamed conv_wino4 kernel -- op tests first, then layer timings, suite, bench
set -o pipefail
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -q -k "winograd4 or groupnorm or conv" > $O/t2_ops.log 2>&1
rc=$?
grep -E "^FAILED|passed|failed" $O/t2_ops.log | tail -30
if [ $rc -ne 0 ]; then
  timeout -k 10 120 python scripts/w4_diag.py 3 2 32 64 28 56 2>&1 | tail -12
  timeout -k 10 120 python scripts/w4_diag.py 3 1 64 64 8 112 pro 2>&1 | tail -12
  timeout -k 10 120 python scripts/w4_diag.py 3 2 32 64 32 64 2>&1 | tail -12
  exit 1
fi
timeout -k 10 300 python scripts/conv_bench.py --only 3x3 --algos 3,4 > $O/cb2.log 2>&1 || { echo "conv_bench FAILED"; tail -20 $O/cb2.log; exit 1; }
cat $O/cb2.log | tail -30
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/t2_full.log 2>&1 || { echo "full FAILED"; tail -40 $O/t2_full.log; exit 1; }
tail -2 $O/t2_full.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train-leg > $O/b2.json 2> $O/b2.err && python3 -c "
import json; d=json.load(open('$O/b2.json')); print(d['value'], d['ms_per_step'], d['launches_per_step'], [(k['kernel'][:18], k['frac'], k['ms_per_step_single_stream'], k['avg_launch_ms']) for k in d['roofline']['kernels']])"
