#!/bin/bash
# split-lane merge kernel of the cross-attention: tests, step table, bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04/r28; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops2_gpu.py tests/test_attn_pin_gpu.py tests/test_configs_gpu.py tests/test_train_gpu.py -x -q -m gpu > $O/t28.log 2>&1; rc=$?; tail -3 $O/t28.log; [ $rc -eq 0 ] || exit 1
IDIFF_HIP_GRAPH=0 IDIFF_TWO_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s1 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-train-leg > $O/s1.log 2>&1
python3 scripts/step_trace.py $O/s1 $O/g.csv > $O/g.txt; rm -rf $O/s1; head -3 $O/g.txt; grep -E "combine|xattn_w" $O/g.txt
for i in 1 2; do python3 bench.py --no-cpu-baseline --no-roofline --no-train-leg 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
