#!/bin/bash
# r05 run 9: SelectConvFn (fused output layer backward) tests, training line, training profile (two streams)
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run09; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu > $O/t_train.log 2>&1; rc=$?; tail -3 $O/t_train.log; [ $rc -eq 0 ] || { tail -60 $O/t_train.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "train or grad or B32 or step" > $O/t_cfg.log 2>&1; tail -3 $O/t_cfg.log
bash scripts/train_profile.sh $O/train 32 > $O/train_profile.txt 2>&1; rm -rf $O/train/stats; head -60 $O/train_profile.txt
python3 bench.py --mode train --batch 32 --steps 5 --warmup 2 > $O/train.json 2> $O/train.err; grep -o '"roofline".*' $O/train.json | cut -c1-3000
