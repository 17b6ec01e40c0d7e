#!/bin/bash
# r05 run 10: stacked token chains of the training step: unit + whole-model gradient tests, training line A/B
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run10; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu > $O/t_train.log 2>&1; rc=$?; tail -3 $O/t_train.log; [ $rc -eq 0 ] || { tail -80 $O/t_train.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "train or grad or B32 or step" > $O/t_cfg.log 2>&1; rc=$?; tail -3 $O/t_cfg.log; [ $rc -eq 0 ] || { tail -60 $O/t_cfg.log; exit 1; }
for v in 0 1 0 1; do
  IDIFF_TRAIN_STACKED=$v python3 bench.py --mode train --batch 32 --steps 5 --warmup 2 --no-roofline > $O/train_stacked$v.json 2> $O/train_stacked$v.err; echo "[IDIFF_TRAIN_STACKED=$v] $(cut -c1-210 $O/train_stacked$v.json) $(grep -o '"library_launches_per_it": [0-9.]*' $O/train_stacked$v.json)"
done
python3 bench.py --mode train --batch 32 --steps 5 --warmup 2 > $O/train.json 2> $O/train.err; grep -o '"roofline".*' $O/train.json | cut -c1-2500
