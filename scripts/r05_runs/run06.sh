#!/bin/bash
# r05 run 6: register-resident compact memory projection: op tests, A/B against the r04 form is the previous call's default line on
# other boxes, so the steady-state step table is taken here
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run06; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ops2_gpu.py tests/test_attn_pin_gpu.py -x -q -m gpu > $O/t_ops2.log 2>&1; rc=$?; tail -3 $O/t_ops2.log; [ $rc -eq 0 ] || { tail -40 $O/t_ops2.log; exit 1; }
for i in 1 2; do python3 bench.py --no-cpu-baseline --no-roofline --no-train-leg 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/[default] /"; done
IDIFF_HIP_GRAPH=0 IDIFF_TWO_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ss -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-train-leg > $O/ss.log 2>&1
python3 scripts/step_trace.py $O/ss $O/step.csv > $O/step.txt; rm -rf $O/ss; head -40 $O/step.txt
timeout -k 10 900 python -m pytest tests/test_configs_gpu.py tests/test_sampling_gpu.py tests/test_unet_gpu.py -x -q -m gpu > $O/t_chain.log 2>&1; tail -3 $O/t_chain.log
