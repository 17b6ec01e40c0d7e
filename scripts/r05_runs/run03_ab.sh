#!/bin/bash
# r05 run 3: grouped ScoreMapModule launches A/B (grouped kernel held to 2 waves/SIMD), side streams as first-level forks (once, under
# faulthandler), training path with fork / skip buffers: its tests, then the training line
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_ops2_gpu.py -x -q -m gpu > $O/t_train.log 2>&1; rc=$?; tail -3 $O/t_train.log; [ $rc -eq 0 ] || { tail -40 $O/t_train.log; exit 1; }
for i in 1 2; do
  IDIFF_GROUPED_SMM=0 python3 bench.py --no-cpu-baseline --no-roofline --no-train-leg 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/[single launches] /"
  python3 bench.py --no-cpu-baseline --no-roofline --no-train-leg 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/[grouped] /"
done
IDIFF_SMM_SIDE=1 timeout -k 10 120 python3 -X faulthandler bench.py --no-cpu-baseline --no-roofline --no-train-leg > $O/side.json 2> $O/side.err; echo "side rc=$?"; grep -o '"ms_per_step": [0-9.]*' $O/side.json | sed "s/^/[IDIFF_SMM_SIDE=1 first-level forks] /"; tail -12 $O/side.err
python3 bench.py --mode train --batch 32 --steps 5 --warmup 2 > $O/train.json 2> $O/train.err; cut -c1-1500 $O/train.json
timeout -k 10 600 python -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "train or grad or B32 or step" > $O/t_cfg.log 2>&1; tail -3 $O/t_cfg.log
