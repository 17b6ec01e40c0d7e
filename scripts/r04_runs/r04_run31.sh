#!/bin/bash
# young-wait immediate counts the 4 statistics stores too (NL+20): parity tests, per-layer A/B against the committed library, step A/B
O=gpurun_out/r04/r31; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "winograd4 or wino4 or conv" > $O/t31.log 2>&1; rc=$?; tail -2 $O/t31.log; [ $rc -eq 0 ] || exit 1
for lib in base y20 base y20; do
  echo "== $lib"
  IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$lib.so timeout -k 10 200 python scripts/conv_bench.py --rounds 5 --iters 5 --only 3x3 2>&1 | grep -E "L0 64|L0up|L1 64|L2up|up 128" | cut -c1-150
done
for lib in base y20 base y20; do
  IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$lib.so python3 bench.py --no-cpu-baseline --no-roofline --no-train-leg 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$lib /"
done
