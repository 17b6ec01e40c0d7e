#!/bin/bash
# wgrad reduce rewrite: training tests, host-issue probe, train line
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_train_gpu.py -x -q -m gpu > $O/t15_train.log 2>&1; echo "train tests rc=$?"; tail -3 $O/t15_train.log
timeout -k 10 300 python scripts/train_issue_probe.py > $O/p15_issue.log 2>&1 && cat $O/p15_issue.log | tail -6 &&
timeout -k 10 300 python bench.py --mode train --batch 32 --steps 6 --warmup 2 > $O/e15_train.json 2> $O/e15_train.err && cat $O/e15_train.json | cut -c1-260
