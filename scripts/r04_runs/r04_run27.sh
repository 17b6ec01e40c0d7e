#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04/ov27; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace2 -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-train-leg > $O/trace2.log 2>&1
python3 scripts/step_overlap.py $O/trace2 > $O/g_two_stream_step_overlap.txt; cat $O/g_two_stream_step_overlap.txt; rm -rf $O/trace2
