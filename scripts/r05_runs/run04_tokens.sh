#!/bin/bash
# r05 run 4: token-side fused Functions of the training path (HeadFoldFn / Linear3Fn / TokenAttnFn / forks): training tests, training line,
# the ATen rows left; the capture fork/join probe
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run04; mkdir -p $O
timeout -k 10 300 python3 scripts/proto/capture_fork_probe.py > $O/capture_fork_probe.txt 2>&1; cat $O/capture_fork_probe.txt
timeout -k 10 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu > $O/t_train.log 2>&1; rc=$?; tail -3 $O/t_train.log; [ $rc -eq 0 ] || { tail -60 $O/t_train.log; exit 1; }
python3 bench.py --mode train --batch 32 --steps 5 --warmup 2 > $O/train.json 2> $O/train.err; cut -c1-400 $O/train.json; grep -o '"device_launches_per_it.*' $O/train.json | cut -c1-900
timeout -k 10 600 python -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "train or grad or B32 or step" > $O/t_cfg.log 2>&1; tail -3 $O/t_cfg.log
python3 scripts/train_aten_sources.py > $O/aten.log 2>&1; grep -v Warning $O/aten.log | head -50 | cut -c1-200
