#!/bin/bash
set -o pipefail
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_rccl_gpu.py -x -q > $O/t9_train.log 2>&1 || { echo "train tests FAILED"; grep -E "^FAILED|Error|assert" $O/t9_train.log | tail -20; tail -30 $O/t9_train.log; exit 1; }
tail -1 $O/t9_train.log
timeout -k 10 400 python bench.py --mode train --batch 32 --steps 6 --warmup 2 > $O/e9_train.json 2> $O/e9_train.err; python3 -c "
import json; d=json.load(open('$O/e9_train.json')); print('train', d['value'], d['ms_per_step'])"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/t9_full.log 2>&1 || { echo "full FAILED"; tail -40 $O/t9_full.log; exit 1; }
tail -2 $O/t9_full.log
