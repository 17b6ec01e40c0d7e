#!/bin/bash
# r05 run 7: compact (72-row) memory in the training step: CompactMemFn / SmmXattnFn(Cm = 72) unit tests, whole-model gradient parity,
# training line A/B (IDIFF_TRAIN_COMPACT=0 = the r04 256-row path); time MLP kernel v2
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run07; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_ops2_gpu.py -x -q -m gpu > $O/t_train.log 2>&1; rc=$?; tail -3 $O/t_train.log; grep "rel err" $O/t_train.log | tail -30; [ $rc -eq 0 ] || { tail -60 $O/t_train.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "train or grad or B32 or step" > $O/t_cfg.log 2>&1; tail -3 $O/t_cfg.log
for v in 0 1; do
  IDIFF_TRAIN_COMPACT=$v python3 bench.py --mode train --batch 32 --steps 5 --warmup 2 --no-roofline > $O/train_compact$v.json 2> $O/train_compact$v.err; echo "[IDIFF_TRAIN_COMPACT=$v] $(cut -c1-330 $O/train_compact$v.json)"
done
python3 bench.py --no-cpu-baseline --no-roofline --no-train-leg 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/[sampling default] /"
