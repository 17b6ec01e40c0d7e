#!/bin/bash
set -o pipefail
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_train_gpu.py -q -x > $O/t13_train.log 2>&1 || { grep -E "^FAILED|passed|failed|Error" $O/t13_train.log | tail; exit 1; }
tail -1 $O/t13_train.log
bash scripts/train_profile.sh $O/train13 32 > $O/train13_profile.txt 2>&1; head -24 $O/train13_profile.txt | cut -c1-125
