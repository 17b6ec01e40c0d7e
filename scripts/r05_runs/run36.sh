#!/bin/bash
# conv1x1_x3_kernel: residual / aux rows of the epilogue requested up front (x3pre) against the tree and a control rebuild (x3ctl)
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run36; mkdir -p $O
IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_x3pre.so timeout -k 10 400 python3 -m pytest tests/test_ops_gpu.py tests/test_ops2_gpu.py -x -q -k "1x1 or x3 or resblock or conv" 2>&1 | tail -1 | tee $O/tests.txt
for r in 1 2; do for lib in tree x3ctl x3pre; do
  if [ $lib = tree ]; then unset IDIFF_LIB; else export IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$lib.so; fi
  echo "== $lib"; python3 scripts/conv_bench.py --only "1x1" --rounds 4 --iters 5 2>&1 | grep " us " | cut -c1-70
  echo -n "bench: "; python3 bench.py --no-cpu-baseline --no-train-leg --no-roofline 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'steps/s', d['ms_per_step'], 'ms/step')"
done; done 2>&1 | tee $O/ab.txt
