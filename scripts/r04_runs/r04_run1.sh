#!/bin/bash
# round 4, GPU run 1: the fused GroupNorm tail after the sign-extension fix (cold process first), then the whole suite with the
# tail on everywhere, then the default suite, then a bench A/B
set -o pipefail
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -s -k "groupnorm" > $O/t_gn_cold.log 2>&1 || { echo "gn cold FAILED"; tail -30 $O/t_gn_cold.log; exit 1; }
tail -3 $O/t_gn_cold.log
IDIFF_GN_FUSED=1 timeout -k 10 900 python -m pytest tests -x -q -m gpu -s > $O/t_full_gnfused.log 2>&1 || { echo "full fused FAILED"; tail -40 $O/t_full_gnfused.log; exit 1; }
tail -2 $O/t_full_gnfused.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/t_full_default.log 2>&1 || { echo "full default FAILED"; tail -40 $O/t_full_default.log; exit 1; }
tail -2 $O/t_full_default.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/ab_default.json 2> $O/ab_default.err && tail -c 600 $O/ab_default.json
IDIFF_GN_FUSED=1 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/ab_gnfused.json 2> $O/ab_gnfused.err && tail -c 600 $O/ab_gnfused.json
