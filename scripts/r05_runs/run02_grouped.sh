#!/bin/bash
# r05 run 2: grouped ScoreMapModule launches + fused time MLP: new op tests, the whole GPU suite, headline A/B, the side-stream
# experiment once under faulthandler, the 2-rank rehearsal of the default line (training leg across ranks)
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run02; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ops2_gpu.py -x -q -m gpu > $O/t_ops2.log 2>&1; tail -3 $O/t_ops2.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/t_full.log 2>&1; rc=$?; tail -3 $O/t_full.log; [ $rc -eq 0 ] || exit 1
python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json; grep -o '"launches_per_step": [0-9.]*' $O/bench.json
grep -o '"train": {.*' $O/bench.json | cut -c1-1800
for i in 1 2; do
  python3 bench.py --no-cpu-baseline --no-roofline --no-train-leg 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/[default] /"
  IDIFF_SMM_SIDE=1 timeout -k 10 120 python3 -X faulthandler bench.py --no-cpu-baseline --no-roofline --no-train-leg > $O/side_$i.json 2> $O/side_$i.err; echo "side rc=$?"; grep -o '"ms_per_step": [0-9.]*' $O/side_$i.json | sed "s/^/[IDIFF_SMM_SIDE=1] /"; tail -5 $O/side_$i.err
done
IDIFF_BENCH_REHEARSAL=1 timeout -k 10 600 python3 bench.py --gpus 2 --no-cpu-baseline --no-roofline > $O/rehearsal_gpus2.json 2> $O/rehearsal_gpus2.err; echo "rehearsal rc=$?"; tail -3 $O/rehearsal_gpus2.err; grep -o '"train": {.*' $O/rehearsal_gpus2.json | cut -c1-2500
