#!/bin/bash
# XCD-aware tile order of the final conv (conv3x3_select): parity, per-kernel time in the step table, step A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04/r33; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ops2_gpu.py tests/test_unet_gpu.py -x -q -m gpu > $O/t33.log 2>&1; rc=$?; tail -2 $O/t33.log; [ $rc -eq 0 ] || exit 1
for lib in base selx; do
IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$lib.so IDIFF_HIP_GRAPH=0 IDIFF_TWO_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s_$lib -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-train-leg > $O/s_$lib.log 2>&1
python3 scripts/step_trace.py $O/s_$lib > $O/g_$lib.txt; rm -rf $O/s_$lib; echo "== $lib"; head -2 $O/g_$lib.txt; grep -E "select" $O/g_$lib.txt
done
for lib in base selx base selx; do
  IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$lib.so python3 bench.py --no-cpu-baseline --no-roofline --no-train-leg 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$lib /"
done
