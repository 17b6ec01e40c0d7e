#!/bin/bash
# r05 run 1: the r04 tree on this round's box -- headline without the CPU leg, the single-stream steady-state step, the training line
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run01; mkdir -p $O
python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
IDIFF_HIP_GRAPH=0 IDIFF_TWO_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ss -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-train-leg > $O/ss.log 2>&1
python3 scripts/step_trace.py $O/ss $O/step.csv > $O/step.txt; rm -rf $O/ss; head -40 $O/step.txt
bash scripts/train_profile.sh $O/train 32 > $O/train_profile.txt 2>&1; rm -rf $O/train/stats; head -45 $O/train_profile.txt
