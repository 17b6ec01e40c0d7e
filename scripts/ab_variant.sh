set -e
V=$1
export IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$V.so
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -k "winograd4" > gpurun_out/ab_${V}_tests.log 2>&1 || { tail -20 gpurun_out/ab_${V}_tests.log; exit 1; }
tail -2 gpurun_out/ab_${V}_tests.log
for lib in base $V; do
  echo "== $lib"
  IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$lib.so timeout -k 10 200 python scripts/conv_bench.py --rounds 5 --iters 5 --only 3x3 2>&1 | tee gpurun_out/ab_${lib}_bench.log
done
