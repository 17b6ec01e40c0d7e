#!/bin/bash
# r05 run 5: capture probe (three more shapes), side-stream experiment rebuilt on the working shape (decoder continues on the side stream),
# thin weight-gradient kernel for Cout <= 16 / Cin <= 8, training tests + line, sampling tests
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run05; mkdir -p $O
timeout -k 10 300 python3 scripts/proto/capture_fork_probe.py > $O/capture_fork_probe.txt 2>&1; cat $O/capture_fork_probe.txt
for i in 1 2; do
  python3 bench.py --no-cpu-baseline --no-roofline --no-train-leg 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/[default] /"
  IDIFF_SMM_SIDE=1 timeout -k 10 120 python3 -X faulthandler bench.py --no-cpu-baseline --no-roofline --no-train-leg > $O/side_$i.json 2> $O/side_$i.err; echo "side rc=$?"; grep -o '"ms_per_step": [0-9.]*' $O/side_$i.json | sed "s/^/[IDIFF_SMM_SIDE=1] /"; grep -A8 "Fatal" $O/side_$i.err | head -12
done
IDIFF_SMM_SIDE=1 timeout -k 10 300 python -m pytest tests/test_sampling_gpu.py -x -q -m gpu -k "chain or graph" > $O/t_side_sampling.log 2>&1; tail -3 $O/t_side_sampling.log
timeout -k 10 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu > $O/t_train.log 2>&1; rc=$?; tail -3 $O/t_train.log; [ $rc -eq 0 ] || { tail -60 $O/t_train.log; exit 1; }
python3 bench.py --mode train --batch 32 --steps 5 --warmup 2 > $O/train.json 2> $O/train.err; cut -c1-400 $O/train.json; grep -o '"aten_launches_per_it.*' $O/train.json | cut -c1-900
bash scripts/train_profile.sh $O/train 32 > $O/train_profile.txt 2>&1; rm -rf $O/train/stats; head -50 $O/train_profile.txt
